/* _ctd_pyblocks: builds the reference's result records -- `TextBlock` objects (reference utils/textblock.py:12-98) -- from the
 * native tail's per-block columns in ONE C loop (CPython API).  `textblock.blocks_from_records` did this in Python: 3.7 us
 * per block, i.e. 3.5 ms of interpreter-lock time per batch of 32 pages at 30 blocks a page and 8 ms on crowded pages, on
 * worker threads that share the lock with the thread launching the forwards.  Here a block costs ~1 us: the attribute dict
 * is a copy of a template that already holds every key in the reference's attribute order (`to_dict()` dumps the dict in
 * creation order, so the order is part of the contract) with the detection fields overwritten.
 *
 * This is marshalling, not arithmetic: every value placed in a record was computed by the native tail (csrc/host_group.cpp)
 * or by numpy on its columns (`distance`, textblock.py:327-328) before the call; the objects are identical to what the
 * Python loop built (tests/test_textblock_helpers.py, tests/test_group_native.py compare them field by field).
 */
#define PY_SSIZE_T_CLEAN
#include <Python.h>

static PyObject *k_copy;
static PyObject *k_xyxy, *k_lines, *k_vertical, *k_language, *k_font_size, *k_distance, *k_angle, *k_vec, *k_norm, *k_merged,
    *k_weight, *k_text;

static int need_list(PyObject* o, Py_ssize_t n, const char* what) {
  if (!PyList_Check(o) || PyList_GET_SIZE(o) < n) {
    PyErr_Format(PyExc_TypeError, "build_blocks: `%s` must be a list of at least %zd entries", what, n);
    return -1;
  }
  return 0;
}

/* build_blocks(cls, template, n, xyxy, all_lines, line_off, n_lines, vertical, language, langs, font, font_is_float,
 *              dval, dist_off, n_dist, angle, vec, norm, merged, weight) -> list of n `cls` instances */
static PyObject* build_blocks(PyObject* self, PyObject* args) {
  PyObject *cls, *tmpl, *xyxy, *all_lines, *line_off, *n_lines, *vertical, *language, *langs, *font, *isf, *dval, *dist_off,
      *n_dist, *angle, *vec, *norm, *merged, *weight;
  Py_ssize_t n;
  if (!PyArg_ParseTuple(args, "OO!nOOOOOOOOOOOOOOOOO", &cls, &PyDict_Type, &tmpl, &n, &xyxy, &all_lines, &line_off, &n_lines,
                        &vertical, &language, &langs, &font, &isf, &dval, &dist_off, &n_dist, &angle, &vec, &norm, &merged,
                        &weight))
    return NULL;
  if (!PyType_Check(cls)) {
    PyErr_SetString(PyExc_TypeError, "build_blocks: `cls` must be a class");
    return NULL;
  }
  if (need_list(xyxy, n, "xyxy") || need_list(all_lines, 0, "all_lines") || need_list(line_off, n, "line_off") ||
      need_list(n_lines, n, "n_lines") || need_list(vertical, n, "vertical") || need_list(language, n, "language") ||
      need_list(font, n, "font") || need_list(isf, n, "font_is_float") || need_list(dist_off, n, "dist_off") ||
      need_list(n_dist, n, "n_dist") || need_list(angle, n, "angle") || need_list(vec, n, "vec") ||
      need_list(norm, n, "norm") || need_list(merged, n, "merged") || need_list(weight, n, "weight"))
    return NULL;
  if (!PySequence_Check(langs)) {
    PyErr_SetString(PyExc_TypeError, "build_blocks: `langs` must be a sequence");
    return NULL;
  }
  PyTypeObject* tp = (PyTypeObject*)cls;
  PyObject* out = PyList_New(n);
  if (!out) return NULL;
  for (Py_ssize_t i = 0; i < n; ++i) {
    PyObject* d = PyDict_Copy(tmpl);
    if (!d) goto fail;
    int rc = 0;
    PyObject* tmp;
    rc |= PyDict_SetItem(d, k_xyxy, PyList_GET_ITEM(xyxy, i));
    {
      const Py_ssize_t lo = PyLong_AsSsize_t(PyList_GET_ITEM(line_off, i)), nl = PyLong_AsSsize_t(PyList_GET_ITEM(n_lines, i));
      tmp = PyList_GetSlice(all_lines, lo, lo + nl);
      if (!tmp) { Py_DECREF(d); goto fail; }
      rc |= PyDict_SetItem(d, k_lines, tmp);
      Py_DECREF(tmp);
    }
    rc |= PyDict_SetItem(d, k_vertical, PyLong_AsLong(PyList_GET_ITEM(vertical, i)) != 0 ? Py_True : Py_False);
    {
      tmp = PySequence_GetItem(langs, PyLong_AsSsize_t(PyList_GET_ITEM(language, i)));
      if (!tmp) { Py_DECREF(d); goto fail; }
      rc |= PyDict_SetItem(d, k_language, tmp);
      Py_DECREF(tmp);
    }
    {
      PyObject* f = PyList_GET_ITEM(font, i);
      if (PyLong_AsLong(PyList_GET_ITEM(isf, i)) != 0) {
        rc |= PyDict_SetItem(d, k_font_size, f);
      } else {                                             /* int(font_size): truncation toward zero */
        tmp = PyLong_FromDouble(PyFloat_AsDouble(f));
        if (!tmp) { Py_DECREF(d); goto fail; }
        rc |= PyDict_SetItem(d, k_font_size, tmp);
        Py_DECREF(tmp);
      }
    }
    {
      const Py_ssize_t lo = PyLong_AsSsize_t(PyList_GET_ITEM(dist_off, i)), nd = PyLong_AsSsize_t(PyList_GET_ITEM(n_dist, i));
      /* the block's OWN array (a copy of the slice): holding one block must not keep the batch's arrays alive, and an
         in-place edit of one block's `distance` / `vec` must not reach another's */
      PyObject* view = PySequence_GetSlice(dval, lo, lo + nd);
      if (!view) { Py_DECREF(d); goto fail; }
      tmp = PyObject_CallMethodNoArgs(view, k_copy);
      Py_DECREF(view);
      if (!tmp) { Py_DECREF(d); goto fail; }
      rc |= PyDict_SetItem(d, k_distance, tmp);
      Py_DECREF(tmp);
    }
    rc |= PyDict_SetItem(d, k_angle, PyList_GET_ITEM(angle, i));
    tmp = PyObject_CallMethodNoArgs(PyList_GET_ITEM(vec, i), k_copy);
    if (!tmp) { Py_DECREF(d); goto fail; }
    rc |= PyDict_SetItem(d, k_vec, tmp);
    Py_DECREF(tmp);
    rc |= PyDict_SetItem(d, k_norm, PyList_GET_ITEM(norm, i));
    rc |= PyDict_SetItem(d, k_merged, PyLong_AsLong(PyList_GET_ITEM(merged, i)) != 0 ? Py_True : Py_False);
    rc |= PyDict_SetItem(d, k_weight, PyList_GET_ITEM(weight, i));
    tmp = PyList_New(0);                                    /* `text`: a fresh list per block */
    if (!tmp) { Py_DECREF(d); goto fail; }
    rc |= PyDict_SetItem(d, k_text, tmp);
    Py_DECREF(tmp);
    if (rc || PyErr_Occurred()) { Py_DECREF(d); goto fail; }
    PyObject* obj = tp->tp_alloc(tp, 0);                    /* `cls.__new__(cls)`: no __init__ */
    if (!obj) { Py_DECREF(d); goto fail; }
    if (PyObject_GenericSetDict(obj, d, NULL) < 0) { Py_DECREF(obj); Py_DECREF(d); goto fail; }
    Py_DECREF(d);
    PyList_SET_ITEM(out, i, obj);
  }
  return out;
fail:
  Py_DECREF(out);
  if (!PyErr_Occurred()) PyErr_SetString(PyExc_RuntimeError, "build_blocks failed");
  return NULL;
}

static PyMethodDef methods[] = {{"build_blocks", build_blocks, METH_VARARGS, "native records -> list of TextBlock objects"},
                                {NULL, NULL, 0, NULL}};
static struct PyModuleDef moddef = {PyModuleDef_HEAD_INIT, "_ctd_pyblocks", "TextBlock construction in C (see pyblocks.c)", -1,
                                    methods};

PyMODINIT_FUNC PyInit__ctd_pyblocks(void) {
  k_xyxy = PyUnicode_InternFromString("xyxy");
  k_lines = PyUnicode_InternFromString("lines");
  k_vertical = PyUnicode_InternFromString("vertical");
  k_language = PyUnicode_InternFromString("language");
  k_font_size = PyUnicode_InternFromString("font_size");
  k_distance = PyUnicode_InternFromString("distance");
  k_copy = PyUnicode_InternFromString("copy");
  k_angle = PyUnicode_InternFromString("angle");
  k_vec = PyUnicode_InternFromString("vec");
  k_norm = PyUnicode_InternFromString("norm");
  k_merged = PyUnicode_InternFromString("merged");
  k_weight = PyUnicode_InternFromString("weight");
  k_text = PyUnicode_InternFromString("text");
  return PyModule_Create(&moddef);
}
