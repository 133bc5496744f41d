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

/* make_matches(cls, rows, sequence, offset, d_start, d_end, d_dist, d_matched) -> list of cls instances */
static PyObject *make_matches(PyObject *self, PyObject *args) {
    PyObject *cls_obj, *rows_obj, *seq, *ds, *de, *dd, *dm;
    long long offset;
    if (!PyArg_ParseTuple(args, "OOOLOOOO", &cls_obj, &rows_obj, &seq, &offset, &ds, &de, &dd, &dm)) return NULL;
    if (!PyType_Check(cls_obj)) {
        PyErr_SetString(PyExc_TypeError, "cls must be a class");
        return NULL;
    }
    PyTypeObject *cls = (PyTypeObject *)cls_obj;
    Py_ssize_t os_, oe, od, om;
    if (slot_offset(ds, cls, &os_) || slot_offset(de, cls, &oe) || slot_offset(dd, cls, &od) || slot_offset(dm, cls, &om)) return NULL;
    Py_buffer view;
    if (PyObject_GetBuffer(rows_obj, &view, PyBUF_SIMPLE) != 0) return NULL;
    if (view.len % (Py_ssize_t)sizeof(fz_row) != 0) {
        PyBuffer_Release(&view);
        PyErr_SetString(PyExc_ValueError, "rows: not an array of 24-byte fz_match records");
        return NULL;
    }
    const Py_ssize_t n = view.len / (Py_ssize_t)sizeof(fz_row);
    const fz_row *rows = (const fz_row *)view.buf;
    PyObject *list = PyList_New(n);
    if (!list) { PyBuffer_Release(&view); return NULL; }
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
        PyObject *m = PySequence_GetSlice(seq, (Py_ssize_t)r.start, (Py_ssize_t)r.end);
        if (!s || !e || !d || !m) {
            Py_XDECREF(s); Py_XDECREF(e); Py_XDECREF(d); Py_XDECREF(m);
            goto fail;
        }
        *(PyObject **)((char *)obj + os_) = s;
        *(PyObject **)((char *)obj + oe) = e;
        *(PyObject **)((char *)obj + od) = d;
        *(PyObject **)((char *)obj + om) = m;
    }
    PyBuffer_Release(&view);
    return list;
fail:
    PyBuffer_Release(&view);
    /* entries not reached yet are NULL: PyList's deallocator skips them */
    Py_DECREF(list);
    return NULL;
}

static PyMethodDef methods[] = {
    {"make_matches", make_matches, METH_VARARGS,
     "make_matches(cls, rows, sequence, offset, d_start, d_end, d_dist, d_matched) -> [cls(start + offset, end + offset, "
     "dist, sequence[start:end]) for the fz_match rows], filled through the slot descriptors"},
    {NULL, NULL, 0, NULL}};

static struct PyModuleDef moduledef = {PyModuleDef_HEAD_INIT, "_fzmatch", "Match objects for fz_match rows", -1, methods};

PyMODINIT_FUNC PyInit__fzmatch(void) { return PyModule_Create(&moduledef); }
