/* Match objects for the rows of an fz_match array (common.py:15-32 of the reference builds one attrs object per
 * match in Python; at 1e3 .. 1e5 results that is the larger part of a find_near_matches call on a resident
 * sequence).  One call fills a list: the instances come from the class's tp_alloc, the four attrs slots are stored
 * through the offsets of their member descriptors (what object.__setattr__ would reach, without the per-attribute
 * calls), `matched` is sequence[start:end].  Rows from the C-ABI satisfy Match's invariants by construction
 * (0 <= start <= end, dist >= 0), so the attrs validators are not run.  Host-side materialisation only — no search
 * arithmetic; fuzzysearch_amd/common.py falls back to a Python loop when this module has not been built. */
#define PY_SSIZE_T_CLEAN
#include <Python.h>
#include <structmember.h>
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

static int slot_offset(PyObject *descr, PyTypeObject *cls, Py_ssize_t *off) {
    if (Py_TYPE(descr) != &PyMemberDescr_Type) {
        PyErr_SetString(PyExc_TypeError, "expected the member descriptors of a __slots__ class");
        return -1;
    }
    PyMemberDescrObject *d = (PyMemberDescrObject *)descr;
    if (d->d_common.d_type != cls || d->d_member->type != T_OBJECT_EX || (d->d_member->flags & READONLY) ||
        d->d_member->offset < (Py_ssize_t)sizeof(PyObject) || d->d_member->offset + (Py_ssize_t)sizeof(PyObject *) > cls->tp_basicsize) {
        PyErr_SetString(PyExc_TypeError, "descriptor does not belong to the class");
        return -1;
    }
    *off = d->d_member->offset;
    return 0;
}

/* rows[0..n) -> list of cls instances.  `matched` = sequence[start:end] — for the exact types bytes and str (what the
 * reference is called with, and what its Match.matched then holds) built directly from the object's storage instead of
 * through a slice object and the mapping protocol (35 instead of 80 ns per match); everything else through
 * PySequence_GetSlice, so that the slice has whatever type the sequence's own slicing gives.  The cyclic collector is
 * switched off while the list is filled: a thousand new container objects would trigger a young-generation pass or two
 * that can find nothing (the objects only reference ints and a slice). */
static PyObject *fill_matches(PyTypeObject *cls, const fz_row *rows, Py_ssize_t n, PyObject *seq, long long offset,
                              Py_ssize_t os_, Py_ssize_t oe, Py_ssize_t od, Py_ssize_t om) {
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
        PyObject *obj = cls->tp_alloc(cls, 0);
        if (!obj) goto fail;
        PyList_SET_ITEM(list, i, obj);                      /* the list owns it from here on (slots are NULL-safe) */
        PyObject *s = PyLong_FromLongLong((long long)r.start + offset);
        PyObject *e = PyLong_FromLongLong((long long)r.end + offset);
        PyObject *d = PyLong_FromLong((long)r.dist);
        PyObject *m;
        if (is_bytes || is_str) {                           /* Python slice semantics: both ends clipped to the length */
            const Py_ssize_t a = r.start < seq_len ? (Py_ssize_t)r.start : seq_len;
            const Py_ssize_t b = r.end < seq_len ? (Py_ssize_t)r.end : seq_len;
            m = is_bytes ? PyBytes_FromStringAndSize(bytes + a, b - a) : PyUnicode_Substring(seq, a, b);
        } else {
            m = PySequence_GetSlice(seq, (Py_ssize_t)r.start, (Py_ssize_t)r.end);
        }
        if (!s || !e || !d || !m) {
            Py_XDECREF(s); Py_XDECREF(e); Py_XDECREF(d); Py_XDECREF(m);
            goto fail;
        }
        *(PyObject **)((char *)obj + os_) = s;
        *(PyObject **)((char *)obj + oe) = e;
        *(PyObject **)((char *)obj + od) = d;
        *(PyObject **)((char *)obj + om) = m;
    }
    FZ_GC_RESUME(gc_was_on);
    return list;
fail:
    FZ_GC_RESUME(gc_was_on);
    /* entries not reached yet are NULL: PyList's deallocator skips them */
    Py_DECREF(list);
    return NULL;
}

static int parse_class(PyObject *cls_obj, PyObject *ds, PyObject *de, PyObject *dd, PyObject *dm, PyTypeObject **cls,
                       Py_ssize_t *os_, Py_ssize_t *oe, Py_ssize_t *od, Py_ssize_t *om) {
    if (!PyType_Check(cls_obj)) {
        PyErr_SetString(PyExc_TypeError, "cls must be a class");
        return -1;
    }
    *cls = (PyTypeObject *)cls_obj;
    return (slot_offset(ds, *cls, os_) || slot_offset(de, *cls, oe) || slot_offset(dd, *cls, od) || slot_offset(dm, *cls, om)) ? -1 : 0;
}

/* make_matches(cls, rows, sequence, offset, d_start, d_end, d_dist, d_matched) -> list of cls instances */
static PyObject *make_matches(PyObject *self, PyObject *args) {
    PyObject *cls_obj, *rows_obj, *seq, *ds, *de, *dd, *dm;
    long long offset;
    if (!PyArg_ParseTuple(args, "OOOLOOOO", &cls_obj, &rows_obj, &seq, &offset, &ds, &de, &dd, &dm)) return NULL;
    PyTypeObject *cls;
    Py_ssize_t os_, oe, od, om;
    if (parse_class(cls_obj, ds, de, dd, dm, &cls, &os_, &oe, &od, &om)) return NULL;
    Py_buffer view;
    if (PyObject_GetBuffer(rows_obj, &view, PyBUF_SIMPLE) != 0) return NULL;
    if (view.len % (Py_ssize_t)sizeof(fz_row) != 0) {
        PyBuffer_Release(&view);
        PyErr_SetString(PyExc_ValueError, "rows: not an array of 24-byte fz_match records");
        return NULL;
    }
    PyObject *list = fill_matches(cls, (const fz_row *)view.buf, view.len / (Py_ssize_t)sizeof(fz_row), seq, offset, os_, oe, od, om);
    PyBuffer_Release(&view);
    return list;
}

/* make_matches_at(cls, address, n, sequence, offset, d_start, d_end, d_dist, d_matched): the same for n fz_match rows at
 * a raw address — the result buffer of a C-ABI call (ctypes hands the pointer over; no numpy array in between). */
static PyObject *make_matches_at(PyObject *self, PyObject *args) {
    PyObject *cls_obj, *seq, *ds, *de, *dd, *dm;
    unsigned long long addr;
    Py_ssize_t n;
    long long offset;
    if (!PyArg_ParseTuple(args, "OKnOLOOOO", &cls_obj, &addr, &n, &seq, &offset, &ds, &de, &dd, &dm)) return NULL;
    PyTypeObject *cls;
    Py_ssize_t os_, oe, od, om;
    if (parse_class(cls_obj, ds, de, dd, dm, &cls, &os_, &oe, &od, &om)) return NULL;
    if (n < 0 || (n > 0 && addr == 0)) {
        PyErr_SetString(PyExc_ValueError, "bad row buffer");
        return NULL;
    }
    return fill_matches(cls, (const fz_row *)(uintptr_t)addr, n, seq, offset, os_, oe, od, om);
}

static PyMethodDef methods[] = {
    {"make_matches", make_matches, METH_VARARGS,
     "make_matches(cls, rows, sequence, offset, d_start, d_end, d_dist, d_matched) -> [cls(start + offset, end + offset, "
     "dist, sequence[start:end]) for the fz_match rows], filled through the slot descriptors"},
    {"make_matches_at", make_matches_at, METH_VARARGS,
     "make_matches_at(cls, address, n, sequence, offset, d_start, d_end, d_dist, d_matched): make_matches for n fz_match rows at a raw address"},
    {NULL, NULL, 0, NULL}};

static struct PyModuleDef moduledef = {PyModuleDef_HEAD_INIT, "_fzmatch", "Match objects for fz_match rows", -1, methods};

PyMODINIT_FUNC PyInit__fzmatch(void) { return PyModule_Create(&moduledef); }
