/*
 * hb_pymarshal.c -- list[int] <-> packed little-endian limbs, in C.
 *
 * The reference boundary converts every element with int.to_bytes / int.from_bytes
 * (hbmpc_ntl_helpers.pyx:20-29, ~0.3 us per element from Python).  This CPython helper does
 * the same conversion ~10x faster so the list-of-int drop-in API is not dominated by
 * marshalling.  Pure plumbing: no field arithmetic except the "reduce on entry" (pyx:31-32),
 * which is delegated to Python's own % operator.
 *
 * It also carries the binding of hb_dec_arrived1 (include/hbmpc_hip.h) as the `add` method of device.DeviceIncrementalDecoder:
 * batch_reconstruct announces one received column per call (reference batch_reconstruction.py:43-61 -> IncrementalDecoder.add,
 * reed_solomon.py:367-403), ~86 calls per open, and a ctypes call costs 0.3-0.4 us where this one costs < 0.1.  Binding only: the
 * decoder's state lives in libhbmpc_hip.so, every state CHANGE is handed to the Python object (`_c_event`).
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

/* ---- DeviceIncrementalDecoder.add ------------------------------------------------------------------------------------------- */
typedef int (*hb_dec_arrived1_fn)(void *, int);
static hb_dec_arrived1_fn g_arrived1 = NULL;
typedef int (*hb_wait_arrived1_fn)(void *, int, int);
static hb_wait_arrived1_fn g_wait1 = NULL;
static PyObject *s_ch, *s_slow, *s_event, *s_wh, *s_wevent, *s_confirmed, *s_avl, *s_zl;

/* bind_dec(address of hb_dec_arrived1) */
static PyObject *hb_bind_dec(PyObject *self, PyObject *arg) {
    void *p = PyLong_AsVoidPtr(arg);
    if (!p && PyErr_Occurred()) return NULL;
    g_arrived1 = (hb_dec_arrived1_fn)p;
    Py_RETURN_NONE;
}

/* bind_wait(address of hb_wait_arrived1) */
static PyObject *hb_bind_wait(PyObject *self, PyObject *arg) {
    void *p = PyLong_AsVoidPtr(arg);
    if (!p && PyErr_Occurred()) return NULL;
    g_wait1 = (hb_wait_arrived1_fn)p;
    Py_RETURN_NONE;
}

/* A decoder whose candidates wait (decoder._wh = the hb_wait handle as an int; columns received in place): the arrival is filtered as
 * IncrementalDecoder.add filters it (reed_solomon.py:369-372: a sender already counted or confirmed in error is ignored), judged by
 * hb_wait_arrived1 with the GIL released, appended to the decoder's arrival list; only an event goes to decoder._w_event(state, idx).
 * Returns NULL with an error set, Py_None's new reference when handled, or (PyObject *)1 when this path does not apply. */
static PyObject *wait_add(PyObject *dec, PyObject *idxobj) {
    if (!g_wait1) return (PyObject *)1;
    PyObject *h = PyObject_GetAttr(dec, s_wh);
    if (!h) return NULL;
    if (h == Py_None) { Py_DECREF(h); return (PyObject *)1; }
    void *p = PyLong_AsVoidPtr(h);
    Py_DECREF(h);
    if (!p && PyErr_Occurred()) return NULL;
    long idx = PyLong_AsLong(idxobj);
    if (idx == -1 && PyErr_Occurred()) return NULL;
    PyObject *conf = PyObject_GetAttr(dec, s_confirmed), *avl = NULL, *zl = NULL, *ret = NULL;
    if (!conf) return NULL;
    avl = PyObject_GetAttr(dec, s_avl);
    zl = avl ? PyObject_GetAttr(dec, s_zl) : NULL;
    if (!avl || !zl) goto out;
    if (!PyAnySet_Check(conf) || !PyAnySet_Check(avl) || !PyList_Check(zl) || idx < 0 || idx > 2147483647L) { ret = (PyObject *)1; goto out; }
    {
        int in = PySet_Contains(conf, idxobj);
        if (in < 0) goto out;
        if (!in) { in = PySet_Contains(avl, idxobj); if (in < 0) goto out; }
        if (in) { Py_INCREF(Py_None); ret = Py_None; goto out; }
        const int nconf = (int)PySet_GET_SIZE(conf);
        int st;
        Py_BEGIN_ALLOW_THREADS
        st = g_wait1(p, (int)idx, nconf);
        Py_END_ALLOW_THREADS
        if (st >= 0 && (PySet_Add(avl, idxobj) < 0 || PyList_Append(zl, idxobj) < 0)) goto out;
        if (st == 0) { Py_INCREF(Py_None); ret = Py_None; goto out; }
        PyObject *sto = PyLong_FromLong(st);
        if (!sto) goto out;
        ret = PyObject_CallMethodObjArgs(dec, s_wevent, sto, idxobj, NULL);
        Py_DECREF(sto);
    }
out:
    Py_XDECREF(conf); Py_XDECREF(avl); Py_XDECREF(zl);
    return ret;
}

/* dec_add(decoder, idx, column=None): while the decoder's optimistic phase lives in C (decoder._ch = the hb_dec handle as an int) and the
 * column was received in place, one call of hb_dec_arrived1 with the GIL released; a state other than "collecting" goes to
 * decoder._c_event(state, idx).  Everything else (a column to copy, keyword arguments, no C decoder) is decoder._add_slow(...). */
static PyObject *hb_dec_add(PyObject *self, PyObject *const *args, Py_ssize_t nargs, PyObject *kwnames) {
    if (nargs < 1) { PyErr_SetString(PyExc_TypeError, "add() needs a decoder"); return NULL; }
    PyObject *dec = args[0];
    if (nargs == 2 && (!kwnames || PyTuple_GET_SIZE(kwnames) == 0) && g_arrived1) {
        PyObject *h = PyObject_GetAttr(dec, s_ch);
        if (!h) return NULL;
        if (h != Py_None) {
            void *p = PyLong_AsVoidPtr(h);
            Py_DECREF(h);
            if (!p && PyErr_Occurred()) return NULL;
            long idx = PyLong_AsLong(args[1]);
            if (idx == -1 && PyErr_Occurred()) return NULL;
            if (idx < -2147483647L || idx > 2147483647L) idx = -1;       /* out of range for the C call: it answers HB_ERR_BAD_ARG */
            int st;
            Py_BEGIN_ALLOW_THREADS
            st = g_arrived1(p, (int)idx);
            Py_END_ALLOW_THREADS
            if (st == 0) Py_RETURN_NONE;
            PyObject *sto = PyLong_FromLong(st);
            if (!sto) return NULL;
            PyObject *r = PyObject_CallMethodObjArgs(dec, s_event, sto, args[1], NULL);
            Py_DECREF(sto);
            return r;
        }
        Py_DECREF(h);
        PyObject *wr = wait_add(dec, args[1]);
        if (wr != (PyObject *)1) return wr;
    }
    PyObject *slow = PyObject_GetAttr(dec, s_slow);
    if (!slow) return NULL;
    PyObject *r = PyObject_Vectorcall(slow, args + 1, (size_t)(nargs - 1), kwnames);
    Py_DECREF(slow);
    return r;
}

/* as_method(callable) -> an object that binds like a function defined in a class body */
static PyObject *hb_as_method(PyObject *self, PyObject *arg) { return PyInstanceMethod_New(arg); }

static PyMethodDef methods[] = {
    {"bind_dec", hb_bind_dec, METH_O, "bind_dec(address of hb_dec_arrived1)"},
    {"bind_wait", hb_bind_wait, METH_O, "bind_wait(address of hb_wait_arrived1)"},
    {"dec_add", (PyCFunction)(void (*)(void))hb_dec_add, METH_FASTCALL | METH_KEYWORDS, "dec_add(decoder, idx, column=None)"},
    {"as_method", hb_as_method, METH_O, "as_method(callable) -> instancemethod"},
    {"pack", hb_pack, METH_VARARGS, "pack(seq, modulus, nbytes) -> bytes"},
    {"unpack", hb_unpack, METH_VARARGS, "unpack(buffer, nbytes) -> list[int]"},
    {NULL, NULL, 0, NULL},
};
static struct PyModuleDef moddef = {PyModuleDef_HEAD_INIT, "_hbmarshal", "int list <-> packed limbs", -1, methods};
PyMODINIT_FUNC PyInit__hbmarshal(void) {
    s_ch = PyUnicode_InternFromString("_ch");
    s_slow = PyUnicode_InternFromString("_add_slow");
    s_event = PyUnicode_InternFromString("_c_event");
    s_wh = PyUnicode_InternFromString("_wh");
    s_wevent = PyUnicode_InternFromString("_w_event");
    s_confirmed = PyUnicode_InternFromString("_confirmed_errors");
    s_avl = PyUnicode_InternFromString("_avl");
    s_zl = PyUnicode_InternFromString("_zl");
    if (!s_ch || !s_slow || !s_event || !s_wh || !s_wevent || !s_confirmed || !s_avl || !s_zl) return NULL;
    return PyModule_Create(&moddef);
}
