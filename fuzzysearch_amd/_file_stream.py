"""find_near_matches_in_file through the streaming pipeline of libfzhip (fz_stream, include/fzhip.h).

The reference (src/fuzzysearch/__init__.py:129-200) reads ``_chunk_size`` items at a time, keeps the
last ``keep`` items of every chunk in front of the next and searches each chunk as an independent
sequence.  With full reads that geometry is regular:

    binary (:129-171)  chunk j = [j * (C - keep), j * (C - keep) + C)            S = C - keep, post = keep
    text   (:174-200)  chunk 0 = [0, C), chunk j = [j * C - keep, (j + 1) * C)  S = C,        pre  = keep

(clipped to the file; chunk j >= 1 exists iff its first NEW item does).  The stream uploads many chunks
per batch and searches them in one launch with per-chunk clamps; this module feeds it (pread threads in
the library for regular files, ``readinto`` for in-memory files, ``read`` + encoding for text), applies
the per-chunk post-processing the reference's strategy classes do, and builds the Match objects of the
final result only.
"""
import io
import os
import stat

import numpy as np

from . import _native
from .common import Match
from .engine import is_byteslike

MODE_EXACT, MODE_LEV, MODE_SUBS, MODE_GENERIC = 0, 1, 2, 3
BATCH_BYTES = 64 << 20
MAX_STREAM_CHUNK = 256 << 20


class Unsupported(Exception):
    """The library refused the geometry / parameters before anything was read."""


class Plan(object):
    __slots__ = ('mode', 'k', 'limits', 'stride', 'pre', 'post', 'binary', 'chunk_size', 'keep', 'group_best')


def plan(search_class, subsequence, search_params, chunk_size, keep, binary, sequence_file):
    """-> Plan, or None when this call has to take the reference's per-chunk loop."""
    from . import ExactSearch, GenericSearch, LevenshteinSearch, SubstitutionsOnlySearch
    m = len(subsequence)
    max_subs, max_ins, max_dels, max_l = search_params.unpacked
    p = Plan()
    p.limits = (0, 0, 0)
    p.group_best = False
    if search_class is ExactSearch:
        p.mode, p.k = MODE_EXACT, 0
    elif search_class is SubstitutionsOnlySearch:
        k = min(x for x in (max_l, max_subs) if x is not None)
        if k == 0:
            p.mode, p.k = MODE_EXACT, 0
        elif m // (k + 1) >= 3:
            p.mode, p.k = MODE_SUBS, k
            p.group_best = True          # bytes: best of every overlap group per chunk; str: all windows
        else:
            return None
    elif search_class is LevenshteinSearch:
        if max_l == 0 or m // (max_l + 1) < 3:
            return None
        p.mode, p.k = MODE_LEV, max_l
    elif search_class is GenericSearch:
        if max_l == 0 or m // (max_l + 1) < 3:
            return None
        p.mode, p.k = MODE_GENERIC, max_l
        unlimited = 1 << 29
        p.limits = tuple(min(unlimited if x is None else x, 255) for x in (max_subs, max_ins, max_dels))
    else:
        return None
    if chunk_size < 2 * keep + 2 or keep < 0:
        return None
    if chunk_size > MAX_STREAM_CHUNK:
        return None                       # the stream pins two staging buffers of two chunks each: per-chunk path instead
    p.binary, p.chunk_size, p.keep = binary, chunk_size, keep
    if binary:
        if not is_byteslike(subsequence):
            return None
        p.stride, p.pre, p.post = chunk_size - keep, 0, keep
        if not _seekable(sequence_file):
            return None                   # `matched` is read back from the file afterwards
    else:
        if not isinstance(subsequence, str):
            return None
        p.stride, p.pre, p.post = chunk_size, keep, 0
    return p


def _seekable(f):
    try:
        return bool(f.seekable())
    except Exception:
        return False


def _regular_fd(f):
    """File descriptor to pread() from — only for objects whose logical bytes ARE the descriptor's bytes:
    io.FileIO itself, or io.BufferedReader / BufferedRandom directly over one.  Everything else that has a
    fileno() (GzipFile, BZ2File, LZMAFile, decrypting or translating wrappers, subclasses) goes through
    readinto(), as in the reference (__init__.py:140-171)."""
    try:
        raw = f
        if type(f) in (io.BufferedReader, io.BufferedRandom):
            raw = f.raw
        if type(raw) is not io.FileIO:
            return None
        fd = raw.fileno()
        if stat.S_ISREG(os.fstat(fd).st_mode):
            return fd
    except Exception:
        pass
    return None


class _TextEncoder(object):
    """str chunks -> one byte per character with identical comparison results against the pattern: latin-1
    where it applies; characters outside it become a byte value the pattern does not contain; a pattern
    that is not latin-1 itself switches both sides to symbol codes (pattern symbols 1..255, the rest 0)."""

    def __init__(self, subsequence):
        try:
            self.pattern = subsequence.encode('latin-1')
            used = set(self.pattern)
            free = [b for b in range(256) if b not in used]
            self.codes = None
            self.filler = free[0] if free else None
            if self.filler is None:
                raise UnicodeEncodeError('latin-1', subsequence, 0, 1, 'pattern uses every byte value')
        except UnicodeEncodeError:
            pts = np.frombuffer(subsequence.encode('utf-32-le'), dtype=np.uint32)
            self.codes = np.unique(pts)
            if len(self.codes) > 255:
                raise Unsupported('more than 255 distinct symbols')
            self.pattern = self._code(pts)

    def _code(self, pts):
        pos = np.minimum(np.searchsorted(self.codes, pts), len(self.codes) - 1)
        return np.where(self.codes[pos] == pts, pos + 1, 0).astype(np.uint8).tobytes()

    def encode(self, text):
        if self.codes is None:
            try:
                return text.encode('latin-1')
            except UnicodeEncodeError:
                pts = np.frombuffer(text.encode('utf-32-le'), dtype=np.uint32)
                return np.where(pts > 255, self.filler, pts).astype(np.uint8).tobytes()
        return self._code(np.frombuffer(text.encode('utf-32-le'), dtype=np.uint32))


def run(p, search_class, subsequence, f):
    engine = _native.default_engine()
    # the whole stream — open .. finish — owns the engine: a fz_ctx runs one stream OR one search at a time,
    # and the default engine is shared between threads (the lock is re-entrant: the stream's own calls nest)
    with engine._lock:
        return _run_locked(engine, p, search_class, subsequence, f)


def _run_locked(engine, p, search_class, subsequence, f):
    store = None
    if p.binary:
        pattern = bytes(bytearray(subsequence)) if not isinstance(subsequence, (bytes, bytearray)) else subsequence
        encoder = None
    else:
        encoder = _TextEncoder(subsequence)
        pattern = encoder.pattern
    try:
        stream = _native.FileStream(engine, p.mode, pattern, p.limits, p.k, p.stride, p.pre, p.post, BATCH_BYTES)
    except NotImplementedError as exc:
        raise Unsupported(str(exc))
    try:
        if p.binary:
            fd = _regular_fd(f)
            start = f.tell()
            if fd is not None:
                total = stream.read_fd(fd, start)
                f.seek(start + total)
            else:
                total = _feed_readinto(stream, f)

            def fetch(s, e):
                f.seek(start + s)
                return bytearray(f.read(e - s))
            end_pos = start + total
        else:
            store = _TextStore(f)
            _feed_text(stream, f, encoder, store)
            fetch = store.get
            end_pos = None
        raw, seg = stream.finish()
    finally:
        stream.close()
    out = _post_process(p, raw, seg, search_class, not p.binary)
    matches = [Match(s, e, d, matched=fetch(s, e)) for (s, e, d) in out]
    if end_pos is not None:
        f.seek(end_pos)
    if store is not None:
        store.done()
    return matches


def _feed_readinto(stream, f):
    total = 0
    while True:
        view = stream.buffer()
        if len(view) == 0:
            break
        got = 0
        while got < len(view):
            n = f.readinto(view[got:])
            if not n:
                break
            got += n
        last = got < len(view)
        del view
        stream.submit(got, last)
        total += got
        if last:
            break
    return total


def _feed_text(stream, f, encoder, store):
    while True:
        view = stream.buffer()
        room = len(view)
        if room == 0:
            break
        cookie = store.mark()
        # read(n) may return fewer than n characters before the end of the file (codecs.StreamReader, network
        # and custom wrappers): only '' means EOF, as in the reference's loop (__init__.py:174-200)
        parts, have, last = [], 0, False
        while have < room:
            piece = f.read(room - have)
            if not piece:
                last = True
                break
            parts.append(piece)
            have += len(piece)
        text = parts[0] if len(parts) == 1 else ''.join(parts)
        data = encoder.encode(text) if text else b''
        view[:len(data)] = data
        del view
        if text:
            store.add(text, cookie)
        stream.submit(len(data), last)
        if last:
            break


class _TextStore(object):
    """Where `matched` of a text-file result comes from.  The reference needs O(_chunk_size) memory; holding every
    decoded piece until the end would need O(file).  For seekable files only (tell() cookie, length) of every piece
    is kept and the pieces that hold a surviving match are read again at the end (results are sorted: one cached
    piece); files that cannot tell()/seek() keep their pieces (they cannot be read twice)."""

    def __init__(self, f):
        self.f = f
        self.reread = False
        try:
            if f.seekable():
                f.tell()
                self.reread = True
        except Exception:
            self.reread = False
        self.pieces = []          # text, or None when it can be read again
        self.cookies = []
        self.lengths = []
        self.starts = None
        self._cached = (-1, None)
        self.end_cookie = None

    def mark(self):
        if self.reread:
            try:
                return self.f.tell()
            except Exception:                      # e.g. "telling position disabled by next() call"
                self.reread = False
        return None

    def add(self, text, cookie):
        keep = cookie is None or not self.reread
        self.pieces.append(text if keep else None)
        self.cookies.append(cookie)
        self.lengths.append(len(text))

    def _piece(self, i):
        if self.pieces[i] is not None:
            return self.pieces[i]
        if self._cached[0] != i:
            if self.end_cookie is None:
                self.end_cookie = self.f.tell()
            self.f.seek(self.cookies[i])
            parts, have = [], 0
            while have < self.lengths[i]:
                piece = self.f.read(self.lengths[i] - have)
                if not piece:
                    break
                parts.append(piece)
                have += len(piece)
            self._cached = (i, ''.join(parts))
        return self._cached[1]

    def get(self, s, e):
        if self.starts is None:
            self.starts = np.cumsum([0] + self.lengths)
        i = int(np.searchsorted(self.starts, s, side='right')) - 1
        out = []
        while s < e and i < len(self.lengths):
            base = int(self.starts[i])
            out.append(self._piece(i)[s - base:e - base])
            s = base + self.lengths[i]
            i += 1
        return ''.join(out)

    def done(self):
        """Leave the file where the reference leaves it: at its end."""
        if self.end_cookie is not None:
            self.f.seek(self.end_cookie)
        self._cached = (-1, None)


def _post_process(p, raw, seg, search_class, text):
    """Raw stream of the whole file (chunk by chunk) -> (start, end, dist) rows of the final result, i.e. what
    search_class.consolidate_matches(chain(search(chunk) for chunk in chunks)) gives in the reference."""
    if p.mode == MODE_EXACT:
        return [(int(s), int(e), 0) for (s, e) in zip(raw["start"].tolist(), raw["end"].tolist())]
    if p.mode == MODE_SUBS:
        rows = []
        if len(raw):
            cuts = np.flatnonzero(np.diff(seg)) + 1
            for part in np.split(raw, cuts):
                if text:                 # substitutions_only.py:160-167: every window once, sorted by start
                    _starts, first = np.unique(part["start"], return_index=True)
                    part = part[first]
                else:                    # :266-282: best of every overlap group, group-creation order
                    part = _native.group_best_array(part)
                rows.extend((s, e, d) for (s, e, d, _g) in part.tolist())
        return rows                      # SubstitutionsOnlySearch.consolidate_matches is the identity
    best = _native.consolidate_array(raw)              # Levenshtein / generic: one consolidation at the end
    return [(s, e, d) for (s, e, d, _g) in best.tolist()]
