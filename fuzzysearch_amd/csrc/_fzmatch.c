/* The Match record of the drop-in API as a C type, and its bulk constructor from the rows of an fz_match array.
 *
 * common.py:15-32 of the reference declares Match as an attrs class (frozen, slots; eq / hash / order on (start, end,
 * dist), `matched` excluded; validated when __debug__) and builds one per match in Python.  At 1e3 .. 1e5 results that
 * is the larger part of a find_near_matches call on a resident sequence, and most of it is allocation: an attrs slots
 * instance (GC-tracked) + two ints beyond the small-int cache + the `matched` slice, and the same four objects freed
 * again when the list goes.  This type keeps start / end / dist as C integers inside the instance (Python ints are made
 * when somebody reads the attribute), so a match is ONE allocation next to its `matched` slice, and it is only tracked by
 * the cyclic collector when `matched` is something that could hold a reference back (not bytes / str).
 *
 * Behaviour kept from the attrs class (tests/test_host_logic.py holds the two against each other): constructor
 * Match(start, end, dist, matched) by position or keyword with the reference's validation and messages (only when
 * __debug__, as there), frozen (attr.exceptions.FrozenInstanceError on assignment and deletion), == / != / hash / ordering
 * on (start, end, dist) against the same class only, repr, pickling / copying, weak references, __match_args__; common.py
 * attaches the attrs field list (__attrs_attrs__), so attr.fields / attr.evolve / attr.asdict work as on the reference's
 * class.  Host-side materialisation only — no search arithmetic; without this module common.py uses the attrs class and
 * a Python loop. */
#define PY_SSIZE_T_CLEAN
#include <Python.h>
#include <structmember.h>
#include <stddef.h>
#include <stdint.h>

typedef struct { int64_t start, end; int32_t dist, block; } fz_row;     /* fz_match of include/fzhip.h */

/* PyGC_Disable / PyGC_Enable are CPython 3.10+; older interpreters fill the list with the collector left as it is (a
 * young-generation pass or two that finds nothing: slower, not wrong) instead of failing to build. */
#if PY_VERSION_HEX >= 0x030A0000
#define FZ_GC_PAUSE() PyGC_Disable()
#define FZ_GC_RESUME(was_on) do { if (was_on) PyGC_Enable(); } while (0)
#else
#define FZ_GC_PAUSE() 0
#define FZ_GC_RESUME(was_on) do { (void)(was_on); } while (0)
#endif

typedef struct {
    PyObject_HEAD
    long long start, end, dist;
    PyObject *matched;
    PyObject *weaklist;
} MatchObject;

static PyTypeObject *match_type = NULL;                 /* the heap type, created at module init */
static PyObject *frozen_error = NULL;                   /* attr.exceptions.FrozenInstanceError (AttributeError without attrs) */

/* `matched` values that cannot be part of a reference cycle leave the instance untracked */
static inline int needs_tracking(PyObject *matched) {
    return !(PyBytes_CheckExact(matched) || PyUnicode_CheckExact(matched));
}

/* An instance of Match itself: allocated without being handed to the collector (PyType_GenericAlloc would track it).
 * Subclasses (which may add a __dict__ or slots of their own) take their type's tp_alloc: zeroed and tracked. */
static MatchObject *match_alloc(PyTypeObject *type) {
    if (type != match_type) return (MatchObject *)type->tp_alloc(type, 0);
    MatchObject *m = PyObject_GC_New(MatchObject, type);    /* (takes the reference a heap type's instances hold on it) */
    if (m) { m->matched = NULL; m->weaklist = NULL; }
    return m;
}

static int match_traverse(MatchObject *self, visitproc visit, void *arg) {
    Py_VISIT(self->matched);
#if PY_VERSION_HEX >= 0x03090000
    Py_VISIT(Py_TYPE(self));
#endif
    return 0;
}

static int match_clear(MatchObject *self) {
    Py_CLEAR(self->matched);
    return 0;
}

static void match_dealloc(MatchObject *self) {
    PyTypeObject *tp = Py_TYPE(self);
    PyObject_GC_UnTrack(self);                              /* (a no-op for the untracked ones) */
    if (self->weaklist) PyObject_ClearWeakRefs((PyObject *)self);
    Py_XDECREF(self->matched);
    tp->tp_free((PyObject *)self);
    Py_DECREF(tp);
}

/* one of the three integer fields: an int (the reference's isinstance(x, int): bools pass) that fits 64 bits */
static int int_field(PyObject *v, const char *message, long long *out) {
    if (!PyLong_Check(v)) {
        PyErr_SetString(PyExc_ValueError, message);
        return -1;
    }
    *out = PyLong_AsLongLong(v);
    return (*out == -1 && PyErr_Occurred()) ? -1 : 0;
}

static int g_optimize = 0;                              /* sys.flags.optimize, read once at import (g_optimize is deprecated) */

static PyObject *match_new(PyTypeObject *type, PyObject *args, PyObject *kwds) {
    static char *kwlist[] = {"start", "end", "dist", "matched", NULL};
    PyObject *s, *e, *d, *matched;
    if (PyTuple_GET_SIZE(args) == 0 && (!kwds || PyDict_GET_SIZE(kwds) == 0)) {
        /* Match.__new__(Match): what copyreg.__newobj__ calls when a pickle of the attrs class (or of this type's Python
         * fallback, common.py) is loaded — the fields follow through __setstate__ */
        MatchObject *m0 = match_alloc(type);
        if (!m0) return NULL;
        m0->start = m0->end = m0->dist = 0;
        Py_INCREF(Py_None);
        m0->matched = Py_None;
        return (PyObject *)m0;
    }
    if (!PyArg_ParseTupleAndKeywords(args, kwds, "OOOO:Match", kwlist, &s, &e, &d, &matched)) return NULL;
    long long vs, ve, vd;
    /* the reference's __attrs_post_init__ (common.py:21-32), in its order; the range checks only when __debug__ */
    if (int_field(s, "start must be a non-negative integer", &vs)) return NULL;
    if (!g_optimize && vs < 0) { PyErr_SetString(PyExc_ValueError, "start must be a non-negative integer"); return NULL; }
    if (int_field(e, "end must be an integer no smaller than start", &ve)) return NULL;
    if (!g_optimize && ve < vs) { PyErr_SetString(PyExc_ValueError, "end must be an integer no smaller than start"); return NULL; }
    if (int_field(d, "dist must be a non-negative integer", &vd)) return NULL;
    if (!g_optimize && vd < 0) { PyErr_SetString(PyExc_ValueError, "dist must be a non-negative integer"); return NULL; }
    if (!g_optimize && matched == Py_None) { PyErr_SetString(PyExc_ValueError, "matched must be supplied"); return NULL; }
    MatchObject *m = match_alloc(type);
    if (!m) return NULL;
    m->start = vs; m->end = ve; m->dist = vd;
    Py_INCREF(matched);
    m->matched = matched;
    if (type == match_type && needs_tracking(matched)) PyObject_GC_Track((PyObject *)m);
    return (PyObject *)m;
}

static int match_setattro(PyObject *self, PyObject *name, PyObject *value) {
    (void)self; (void)name; (void)value;
    PyErr_SetNone(frozen_error);                            /* assignment and deletion alike, as attrs' frozen classes */
    return -1;
}

static inline int cmp3(const MatchObject *a, const MatchObject *b) {
    if (a->start != b->start) return a->start < b->start ? -1 : 1;
    if (a->end != b->end) return a->end < b->end ? -1 : 1;
    if (a->dist != b->dist) return a->dist < b->dist ? -1 : 1;
    return 0;
}

static PyObject *match_richcompare(PyObject *a, PyObject *b, int op) {
    if (Py_TYPE(a) != Py_TYPE(b)) Py_RETURN_NOTIMPLEMENTED;     /* attrs: other.__class__ is self.__class__ */
    const int c = cmp3((MatchObject *)a, (MatchObject *)b);
    int r;
    switch (op) {
    case Py_EQ: r = c == 0; break;
    case Py_NE: r = c != 0; break;
    case Py_LT: r = c < 0; break;
    case Py_LE: r = c <= 0; break;
    case Py_GT: r = c > 0; break;
    default: r = c >= 0; break;
    }
    if (r) Py_RETURN_TRUE;
    Py_RETURN_FALSE;
}

static Py_hash_t match_hash(MatchObject *self) {
    uint64_t h = 0x9E3779B97F4A7C15ull;                          /* equal (start, end, dist) -> equal hash; `matched` is not part */
    const uint64_t v[3] = {(uint64_t)self->start, (uint64_t)self->end, (uint64_t)self->dist};
    for (int i = 0; i < 3; ++i) {
        h ^= v[i] + 0x9E3779B97F4A7C15ull + (h << 6) + (h >> 2);
        h *= 0xFF51AFD7ED558CCDull;
        h ^= h >> 33;
    }
    Py_hash_t r = (Py_hash_t)h;
    return r == -1 ? -2 : r;
}

static PyObject *match_repr(MatchObject *self) {
    /* attrs: self.__class__.__qualname__.rsplit(">.", 1)[-1] */
    PyObject *qual = PyObject_GetAttrString((PyObject *)Py_TYPE(self), "__qualname__");
    if (!qual) return NULL;
    PyObject *parts = PyObject_CallMethod(qual, "rsplit", "si", ">.", 1);
    Py_DECREF(qual);
    if (!parts) return NULL;
    PyObject *name = PySequence_GetItem(parts, PySequence_Size(parts) - 1);
    Py_DECREF(parts);
    if (!name) return NULL;
    PyObject *r = PyUnicode_FromFormat("%U(start=%lld, end=%lld, dist=%lld, matched=%R)", name, self->start, self->end, self->dist,
                                       self->matched ? self->matched : Py_None);
    Py_DECREF(name);
    return r;
}

static PyObject *match_get_matched(MatchObject *self, void *closure) {
    (void)closure;
    PyObject *m = self->matched ? self->matched : Py_None;
    Py_INCREF(m);
    return m;
}

/* __setstate__: the state of an attrs slots class — a dict {field: value} (attrs >= 22.2) or a tuple in field order */
static PyObject *match_setstate(MatchObject *self, PyObject *state) {
    PyObject *v[4] = {NULL, NULL, NULL, NULL};
    static const char *names[4] = {"start", "end", "dist", "matched"};
    if (PyDict_Check(state)) {
        for (int i = 0; i < 4; ++i) v[i] = PyDict_GetItemString(state, names[i]);
    } else if (PyTuple_Check(state) && PyTuple_GET_SIZE(state) == 4) {
        for (int i = 0; i < 4; ++i) v[i] = PyTuple_GET_ITEM(state, i);
    }
    if (!v[0] || !v[1] || !v[2] || !v[3]) { PyErr_SetString(PyExc_TypeError, "Match.__setstate__: a dict or 4-tuple of start, end, dist, matched"); return NULL; }
    long long vs, ve, vd;
    if (int_field(v[0], "start must be a non-negative integer", &vs) || int_field(v[1], "end must be an integer no smaller than start", &ve) ||
        int_field(v[2], "dist must be a non-negative integer", &vd)) return NULL;
    self->start = vs; self->end = ve; self->dist = vd;
    Py_INCREF(v[3]);
    Py_XSETREF(self->matched, v[3]);
    if (Py_TYPE(self) == match_type && needs_tracking(v[3]) && !PyObject_GC_IsTracked((PyObject *)self)) PyObject_GC_Track((PyObject *)self);
    Py_RETURN_NONE;
}

static PyObject *match_reduce(MatchObject *self, PyObject *ignored) {
    (void)ignored;
    return Py_BuildValue("(O(LLLO))", (PyObject *)Py_TYPE(self), self->start, self->end, self->dist,
                         self->matched ? self->matched : Py_None);
}

static PyMemberDef match_members[] = {
    {"start", T_LONGLONG, offsetof(MatchObject, start), READONLY, "index of the match's first item"},
    {"end", T_LONGLONG, offsetof(MatchObject, end), READONLY, "index one past the match's last item"},
    {"dist", T_LONGLONG, offsetof(MatchObject, dist), READONLY, "distance between the subsequence and the matched items"},
#if PY_VERSION_HEX >= 0x03090000
    {"__weaklistoffset__", T_PYSSIZET, offsetof(MatchObject, weaklist), READONLY, NULL},
#endif
    {NULL, 0, 0, 0, NULL}};

static PyGetSetDef match_getset[] = {
    {"matched", (getter)match_get_matched, NULL, "the matched portion of the sequence", NULL},
    {NULL, NULL, NULL, NULL, NULL}};

static PyMethodDef match_methods[] = {
    {"__reduce__", (PyCFunction)match_reduce, METH_NOARGS, "pickle / copy support: (Match, (start, end, dist, matched))"},
    {"__setstate__", (PyCFunction)match_setstate, METH_O, "loads pickles made by the attrs class / the Python fallback (dict or tuple state)"},
    {NULL, NULL, 0, NULL}};

static PyType_Slot match_slots[] = {
    {Py_tp_doc, (void *)"Match(start, end, dist, matched)\n--\n\n"
                        "A near-match: sequence[start:end] is within dist of the subsequence (the reference's attrs class, "
                        "common.py:15-32; equality, hash and ordering on (start, end, dist))."},
    {Py_tp_new, match_new},
    {Py_tp_dealloc, match_dealloc},
    {Py_tp_traverse, match_traverse},
    {Py_tp_clear, match_clear},
    {Py_tp_setattro, match_setattro},
    {Py_tp_richcompare, match_richcompare},
    {Py_tp_hash, match_hash},
    {Py_tp_repr, match_repr},
    {Py_tp_members, match_members},
    {Py_tp_getset, match_getset},
    {Py_tp_methods, match_methods},
    {0, NULL}};

static PyType_Spec match_spec = {"fuzzysearch_amd.common.Match", sizeof(MatchObject), 0, Py_TPFLAGS_DEFAULT | Py_TPFLAGS_HAVE_GC | Py_TPFLAGS_BASETYPE,
                                 match_slots};

/* rows[0..n) -> list of Match.  `matched` = sequence[start:end] — for the exact types bytes and str (what the reference
 * is called with, and what its Match.matched then holds) built directly from the object's storage instead of through a
 * slice object and the mapping protocol; everything else through PySequence_GetSlice, so that the slice has whatever type
 * the sequence's own slicing gives.  Rows from the C-ABI satisfy Match's invariants by construction (0 <= start <= end,
 * dist >= 0); a row that does not is an error.  The cyclic collector is switched off while the list is filled (the
 * allocations would trigger young-generation passes that can find nothing). */
static PyObject *fill_matches(const fz_row *rows, Py_ssize_t n, PyObject *seq, long long offset) {
    PyObject *list = PyList_New(n);
    if (!list) return NULL;
    const int is_bytes = PyBytes_CheckExact(seq), is_str = PyUnicode_CheckExact(seq);
    const char *bytes = is_bytes ? PyBytes_AS_STRING(seq) : NULL;
    const Py_ssize_t seq_len = is_bytes ? PyBytes_GET_SIZE(seq) : is_str ? PyUnicode_GET_LENGTH(seq) : 0;
    const int gc_was_on = FZ_GC_PAUSE();
    for (Py_ssize_t i = 0; i < n; ++i) {
        const fz_row r = rows[i];
        if (r.start < 0 || r.end < r.start || r.dist < 0) {
            PyErr_SetString(PyExc_ValueError, "row violates 0 <= start <= end, dist >= 0");
            goto fail;
        }
        MatchObject *obj = match_alloc(match_type);
        if (!obj) goto fail;
        PyList_SET_ITEM(list, i, (PyObject *)obj);          /* the list owns it from here on (matched may still be NULL) */
        obj->start = (long long)r.start + offset;
        obj->end = (long long)r.end + offset;
        obj->dist = (long long)r.dist;
        if (is_bytes || is_str) {                           /* Python slice semantics: both ends clipped to the length */
            const Py_ssize_t a = r.start < seq_len ? (Py_ssize_t)r.start : seq_len;
            const Py_ssize_t b = r.end < seq_len ? (Py_ssize_t)r.end : seq_len;
            obj->matched = is_bytes ? PyBytes_FromStringAndSize(bytes + a, b - a) : PyUnicode_Substring(seq, a, b);
            if (!obj->matched) goto fail;
        } else {
            obj->matched = PySequence_GetSlice(seq, (Py_ssize_t)r.start, (Py_ssize_t)r.end);
            if (!obj->matched) goto fail;
            if (needs_tracking(obj->matched)) PyObject_GC_Track((PyObject *)obj);
        }
    }
    FZ_GC_RESUME(gc_was_on);
    return list;
fail:
    FZ_GC_RESUME(gc_was_on);
    /* entries not reached yet are NULL: PyList's deallocator skips them */
    Py_DECREF(list);
    return NULL;
}

/* make_matches(rows, sequence, offset) -> list of Match; rows: a C-contiguous buffer of fz_match records */
static PyObject *make_matches(PyObject *self, PyObject *args) {
    PyObject *rows_obj, *seq;
    long long offset;
    (void)self;
    if (!PyArg_ParseTuple(args, "OOL", &rows_obj, &seq, &offset)) return NULL;
    Py_buffer view;
    if (PyObject_GetBuffer(rows_obj, &view, PyBUF_SIMPLE) != 0) return NULL;
    if (view.len % (Py_ssize_t)sizeof(fz_row) != 0) {
        PyBuffer_Release(&view);
        PyErr_SetString(PyExc_ValueError, "rows: not an array of 24-byte fz_match records");
        return NULL;
    }
    PyObject *list = fill_matches((const fz_row *)view.buf, view.len / (Py_ssize_t)sizeof(fz_row), seq, offset);
    PyBuffer_Release(&view);
    return list;
}

/* make_matches_at(address, n, sequence, offset): the same for n fz_match rows at a raw address — the result buffer of a
 * C-ABI call (ctypes hands the pointer over; no numpy array in between). */
static PyObject *make_matches_at(PyObject *self, PyObject *args) {
    PyObject *seq;
    unsigned long long addr;
    Py_ssize_t n;
    long long offset;
    (void)self;
    if (!PyArg_ParseTuple(args, "KnOL", &addr, &n, &seq, &offset)) return NULL;
    if (n < 0 || (n > 0 && addr == 0)) {
        PyErr_SetString(PyExc_ValueError, "bad row buffer");
        return NULL;
    }
    return fill_matches((const fz_row *)(uintptr_t)addr, n, seq, offset);
}

static PyMethodDef methods[] = {
    {"make_matches", make_matches, METH_VARARGS,
     "make_matches(rows, sequence, offset) -> [Match(start + offset, end + offset, dist, sequence[start:end]) for the fz_match rows]"},
    {"make_matches_at", make_matches_at, METH_VARARGS,
     "make_matches_at(address, n, sequence, offset): make_matches for n fz_match rows at a raw address"},
    {NULL, NULL, 0, NULL}};

static struct PyModuleDef moduledef = {PyModuleDef_HEAD_INIT, "_fzmatch", "Match, and Match objects for fz_match rows", -1, methods};

PyMODINIT_FUNC PyInit__fzmatch(void) {
    PyObject *mod = PyModule_Create(&moduledef);
    if (!mod) return NULL;
    {
        PyObject *flags = PySys_GetObject("flags");                 /* borrowed */
        PyObject *opt = flags ? PyObject_GetAttrString(flags, "optimize") : NULL;
        if (opt) { g_optimize = PyLong_AsLong(opt) > 0; Py_DECREF(opt); }
        PyErr_Clear();
    }
    PyObject *exc_mod = PyImport_ImportModule("attr.exceptions");
    if (exc_mod) {
        frozen_error = PyObject_GetAttrString(exc_mod, "FrozenInstanceError");
        Py_DECREF(exc_mod);
    }
    if (!frozen_error) {                                        /* no attrs: its FrozenInstanceError is an AttributeError */
        PyErr_Clear();
        frozen_error = PyExc_AttributeError;
        Py_INCREF(frozen_error);
    }
    PyObject *type = PyType_FromSpec(&match_spec);
    if (!type) { Py_DECREF(mod); return NULL; }
    match_type = (PyTypeObject *)type;
    PyObject *names = Py_BuildValue("(ssss)", "start", "end", "dist", "matched");
    if (!names || PyDict_SetItemString(match_type->tp_dict, "__match_args__", names) != 0) {
        Py_XDECREF(names); Py_DECREF(type); Py_DECREF(mod);
        return NULL;
    }
    Py_DECREF(names);
    PyType_Modified(match_type);
    Py_INCREF(type);                                            /* one reference for the module attribute, one for match_type */
    if (PyModule_AddObject(mod, "Match", type) != 0) { Py_DECREF(type); Py_DECREF(type); Py_DECREF(mod); return NULL; }
    return mod;
}
