/* _kzg_pyconv -- marshalling between Python ints and the C ABI's little-endian limb arrays.
 *
 * The reference's callers hand fft_ff / KZG.commit plain lists of field elements (fft_ff.py:3,
 * kzg.py:80-97), so at 2^20 coefficients the facade's int <-> limb conversion in pure Python
 * (int.to_bytes per element: 0.4 s per list, 0.25 s back) dwarfs the transform it feeds.  This is
 * the same conversion through the CPython API: ints_to_bytes(seq, nbytes) and
 * bytes_to_ints(buffer, nbytes).  Host-side plumbing only -- no field arithmetic here.
 * Built by kzg_snark_amd/build.py (gcc); kzg_snark_amd/_native.py falls back to the Python
 * form when the module is absent.  */
#define PY_SSIZE_T_CLEAN
#include <Python.h>

/* ints_to_bytes(seq, nbytes) -> bytearray (writable, so numpy can wrap it without a copy): every element as an unsigned little-endian integer of nbytes bytes.
 * Elements that are not ints go through int() (field elements).  Negative or too large: OverflowError, as
 * int.to_bytes raises. */
static PyObject* ints_to_bytes(PyObject* self, PyObject* args) {
  PyObject* seq;
  Py_ssize_t nb;
  if (!PyArg_ParseTuple(args, "On", &seq, &nb)) return NULL;
  if (nb <= 0) { PyErr_SetString(PyExc_ValueError, "nbytes must be positive"); return NULL; }
  PyObject* fast = PySequence_Fast(seq, "ints_to_bytes: expected a sequence");
  if (!fast) return NULL;
  const Py_ssize_t n = PySequence_Fast_GET_SIZE(fast);
  PyObject* out = PyByteArray_FromStringAndSize(NULL, n * nb);
  if (!out) { Py_DECREF(fast); return NULL; }
  unsigned char* p = (unsigned char*)PyByteArray_AS_STRING(out);
  for (Py_ssize_t i = 0; i < n; ++i) {
    PyObject* item = PySequence_Fast_GET_ITEM(fast, i);   /* borrowed */
    PyObject* owned = NULL;
    if (!PyLong_Check(item)) {
      owned = PyNumber_Long(item);
      if (!owned) goto fail;
      item = owned;
    }
    /* CPython 3.13 added a `with_exceptions` argument to this private function (build.py treats the helper as
     * optional: should a later CPython change it again, the build warns and the pure-Python conversion is used) */
#if PY_VERSION_HEX >= 0x030D0000
    const int rc = _PyLong_AsByteArray((PyLongObject*)item, p + i * nb, (size_t)nb, 1 /* little */, 0 /* unsigned */, 1);
#else
    const int rc = _PyLong_AsByteArray((PyLongObject*)item, p + i * nb, (size_t)nb, 1 /* little */, 0 /* unsigned */);
#endif
    Py_XDECREF(owned);
    if (rc < 0) goto fail;
  }
  Py_DECREF(fast);
  return out;
fail:
  Py_DECREF(fast);
  Py_DECREF(out);
  return NULL;
}

/* bytes_to_ints(buffer, nbytes) -> list of ints, one per nbytes-byte little-endian group. */
static PyObject* bytes_to_ints(PyObject* self, PyObject* args) {
  Py_buffer view;
  Py_ssize_t nb;
  if (!PyArg_ParseTuple(args, "y*n", &view, &nb)) return NULL;
  if (nb <= 0 || view.len % nb) {
    PyBuffer_Release(&view);
    PyErr_SetString(PyExc_ValueError, "buffer length is not a multiple of nbytes");
    return NULL;
  }
  const Py_ssize_t n = view.len / nb;
  PyObject* out = PyList_New(n);
  if (!out) { PyBuffer_Release(&view); return NULL; }
  const unsigned char* p = (const unsigned char*)view.buf;
  for (Py_ssize_t i = 0; i < n; ++i) {
    PyObject* v = _PyLong_FromByteArray(p + i * nb, (size_t)nb, 1 /* little */, 0 /* unsigned */);
    if (!v) { Py_DECREF(out); PyBuffer_Release(&view); return NULL; }
    PyList_SET_ITEM(out, i, v);
  }
  PyBuffer_Release(&view);
  return out;
}

static PyMethodDef methods[] = {
    {"ints_to_bytes", ints_to_bytes, METH_VARARGS, "sequence of ints -> little-endian bytes, nbytes each"},
    {"bytes_to_ints", bytes_to_ints, METH_VARARGS, "little-endian bytes -> list of ints, nbytes each"},
    {NULL, NULL, 0, NULL}};

static struct PyModuleDef moddef = {PyModuleDef_HEAD_INIT, "_kzg_pyconv", "int <-> limb marshalling", -1, methods};

PyMODINIT_FUNC PyInit__kzg_pyconv(void) { return PyModule_Create(&moddef); }
