/*
 * hb_pymarshal.c -- list[int] <-> packed little-endian limbs, in C.
 *
 * The reference boundary converts every element with int.to_bytes / int.from_bytes
 * (hbmpc_ntl_helpers.pyx:20-29, ~0.3 us per element from Python).  This CPython helper does
 * the same conversion ~10x faster so the list-of-int drop-in API is not dominated by
 * marshalling.  Pure plumbing: no field arithmetic except the "reduce on entry" (pyx:31-32),
 * which is delegated to Python's own % operator.
 */
#define PY_SSIZE_T_CLEAN
#include <Python.h>
#include <string.h>

/* pack(seq, modulus, nbytes) -> bytes of len(seq) * nbytes */
static PyObject *hb_pack(PyObject *self, PyObject *args) {
    PyObject *seq, *modulus;
    int nbytes;
    if (!PyArg_ParseTuple(args, "OOi", &seq, &modulus, &nbytes)) return NULL;
    PyObject *fast = PySequence_Fast(seq, "expected a list or tuple of ints");
    if (!fast) return NULL;
    Py_ssize_t n = PySequence_Fast_GET_SIZE(fast);
    PyObject *out = PyBytes_FromStringAndSize(NULL, n * (Py_ssize_t)nbytes);
    if (!out) { Py_DECREF(fast); return NULL; }
    unsigned char *buf = (unsigned char *)PyBytes_AS_STRING(out);
    for (Py_ssize_t i = 0; i < n; i++) {
        PyObject *v = PySequence_Fast_GET_ITEM(fast, i);
        if (!PyLong_Check(v)) {
            PyErr_SetString(PyExc_TypeError, "field elements must be Python ints");
            goto fail;
        }
        if (_PyLong_Sign(v) < 0) {
            PyErr_SetString(PyExc_OverflowError, "can't convert negative int to unsigned");
            goto fail;
        }
        PyObject *r = NULL;
        int ge = PyObject_RichCompareBool(v, modulus, Py_GE);
        if (ge < 0) goto fail;
        if (ge) { r = PyNumber_Remainder(v, modulus); if (!r) goto fail; v = r; }
        int rc = _PyLong_AsByteArray((PyLongObject *)v, buf + i * nbytes, (size_t)nbytes, 1, 0);
        Py_XDECREF(r);
        if (rc < 0) goto fail;
    }
    Py_DECREF(fast);
    return out;
fail:
    Py_DECREF(fast);
    Py_DECREF(out);
    return NULL;
}

/* unpack(buffer, nbytes) -> list of ints */
static PyObject *hb_unpack(PyObject *self, PyObject *args) {
    Py_buffer view;
    int nbytes;
    if (!PyArg_ParseTuple(args, "y*i", &view, &nbytes)) return NULL;
    if (nbytes <= 0 || view.len % nbytes) {
        PyBuffer_Release(&view);
        PyErr_SetString(PyExc_ValueError, "buffer length is not a multiple of the element size");
        return NULL;
    }
    Py_ssize_t n = view.len / nbytes;
    PyObject *out = PyList_New(n);
    if (!out) { PyBuffer_Release(&view); return NULL; }
    const unsigned char *buf = (const unsigned char *)view.buf;
    for (Py_ssize_t i = 0; i < n; i++) {
        PyObject *v = _PyLong_FromByteArray(buf + i * nbytes, (size_t)nbytes, 1, 0);
        if (!v) { Py_DECREF(out); PyBuffer_Release(&view); return NULL; }
        PyList_SET_ITEM(out, i, v);
    }
    PyBuffer_Release(&view);
    return out;
}

static PyMethodDef methods[] = {
    {"pack", hb_pack, METH_VARARGS, "pack(seq, modulus, nbytes) -> bytes"},
    {"unpack", hb_unpack, METH_VARARGS, "unpack(buffer, nbytes) -> list[int]"},
    {NULL, NULL, 0, NULL},
};
static struct PyModuleDef moddef = {PyModuleDef_HEAD_INIT, "_hbmarshal", "int list <-> packed limbs", -1, methods};
PyMODINIT_FUNC PyInit__hbmarshal(void) { return PyModule_Create(&moddef); }
