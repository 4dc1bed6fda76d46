/* _emb_fastcall: a thin CPython call shim for the hottest entry points of
 * libembodied_hip.so.
 *
 * The library's boundary is the C ABI in include/embodied_hip.h and the
 * package binds it with ctypes (embodied_amd/_lib.py).  ctypes spends 1-2 us
 * converting the arguments of a 10-12 argument call; a vectorised Driver step
 * makes three such calls in ~35 us of host time.  This module calls the SAME
 * exported functions through their addresses (taken from the ctypes handle),
 * converting arguments itself: Python int / None / objects with the buffer
 * protocol (ctypes arrays) -> 64-bit integer registers, Python float -> float.
 * It links against nothing but libpython and knows three call shapes:
 *
 *   ints(addr, a0 .. aN)            every argument is a pointer or an integer
 *   obs_stack(addr, 11 args)        emb_obs_stack  (two floats at 7, 8)
 *   scan(addr, 10 or 11 args)       emb_scan_gae / emb_scan_lambda (floats at 6, 7)
 *
 * On x86-64 SysV an int32 parameter reads the low half of the 64-bit register
 * or stack slot it is passed in, so integer-class arguments are all passed as
 * uint64_t.  The GIL is released around the call, as ctypes does.
 */
#define PY_SSIZE_T_CLEAN
#include <Python.h>
#include <stdint.h>
#include <string.h>

typedef uint64_t u64;

/* Python int (signed or unsigned 64-bit), None -> 0, or a buffer -> its address. */
static int as_u64(PyObject* o, u64* out) {
  if (PyLong_CheckExact(o)) {
    int overflow = 0;
    long long v = PyLong_AsLongLongAndOverflow(o, &overflow);
    if (!overflow) {
      if (v == -1 && PyErr_Occurred()) return -1;
      *out = (u64)v;
      return 0;
    }
    unsigned long long u = PyLong_AsUnsignedLongLong(o);
    if (u == (unsigned long long)-1 && PyErr_Occurred()) return -1;
    *out = (u64)u;
    return 0;
  }
  if (o == Py_None) {
    *out = 0;
    return 0;
  }
  if (PyBool_Check(o) || PyLong_Check(o)) {
    long long v = PyLong_AsLongLong(o);
    if (v == -1 && PyErr_Occurred()) return -1;
    *out = (u64)v;
    return 0;
  }
  if (PyObject_CheckBuffer(o)) {
    Py_buffer view;
    if (PyObject_GetBuffer(o, &view, PyBUF_SIMPLE) < 0) return -1;
    *out = (u64)(uintptr_t)view.buf;
    PyBuffer_Release(&view);       /* the caller keeps the object alive over the call */
    return 0;
  }
  PyErr_Format(PyExc_TypeError, "fastcall: cannot pass %s as a pointer or integer",
               Py_TYPE(o)->tp_name);
  return -1;
}

static int as_float(PyObject* o, float* out) {
  double v = PyFloat_AsDouble(o);
  if (v == -1.0 && PyErr_Occurred()) return -1;
  *out = (float)v;
  return 0;
}

#define MAX_INTS 14

static PyObject* call_ints(PyObject* self, PyObject* const* args, Py_ssize_t nargs) {
  if (nargs < 1 || nargs > MAX_INTS + 1) {
    PyErr_SetString(PyExc_TypeError, "fastcall.ints(addr, up to 14 arguments)");
    return NULL;
  }
  u64 a[MAX_INTS + 1] = {0};
  for (Py_ssize_t i = 0; i < nargs; ++i)
    if (as_u64(args[i], &a[i]) < 0) return NULL;
  void* fn = (void*)(uintptr_t)a[0];
  int32_t status;
  const int n = (int)nargs - 1;
  Py_BEGIN_ALLOW_THREADS
  switch (n) {      /* the exact arity: no reliance on callee ignoring extras */
    case 0: status = ((int32_t(*)(void))fn)(); break;
    case 1: status = ((int32_t(*)(u64))fn)(a[1]); break;
    case 2: status = ((int32_t(*)(u64, u64))fn)(a[1], a[2]); break;
    case 3: status = ((int32_t(*)(u64, u64, u64))fn)(a[1], a[2], a[3]); break;
    case 4: status = ((int32_t(*)(u64, u64, u64, u64))fn)(a[1], a[2], a[3], a[4]); break;
    case 5: status = ((int32_t(*)(u64, u64, u64, u64, u64))fn)(a[1], a[2], a[3], a[4], a[5]); break;
    case 6: status = ((int32_t(*)(u64, u64, u64, u64, u64, u64))fn)(a[1], a[2], a[3], a[4], a[5], a[6]); break;
    case 7: status = ((int32_t(*)(u64, u64, u64, u64, u64, u64, u64))fn)(a[1], a[2], a[3], a[4], a[5], a[6], a[7]); break;
    case 8: status = ((int32_t(*)(u64, u64, u64, u64, u64, u64, u64, u64))fn)(a[1], a[2], a[3], a[4], a[5], a[6], a[7], a[8]); break;
    case 9: status = ((int32_t(*)(u64, u64, u64, u64, u64, u64, u64, u64, u64))fn)(a[1], a[2], a[3], a[4], a[5], a[6], a[7], a[8], a[9]); break;
    case 10: status = ((int32_t(*)(u64, u64, u64, u64, u64, u64, u64, u64, u64, u64))fn)(a[1], a[2], a[3], a[4], a[5], a[6], a[7], a[8], a[9], a[10]); break;
    case 11: status = ((int32_t(*)(u64, u64, u64, u64, u64, u64, u64, u64, u64, u64, u64))fn)(a[1], a[2], a[3], a[4], a[5], a[6], a[7], a[8], a[9], a[10], a[11]); break;
    case 12: status = ((int32_t(*)(u64, u64, u64, u64, u64, u64, u64, u64, u64, u64, u64, u64))fn)(a[1], a[2], a[3], a[4], a[5], a[6], a[7], a[8], a[9], a[10], a[11], a[12]); break;
    case 13: status = ((int32_t(*)(u64, u64, u64, u64, u64, u64, u64, u64, u64, u64, u64, u64, u64))fn)(a[1], a[2], a[3], a[4], a[5], a[6], a[7], a[8], a[9], a[10], a[11], a[12], a[13]); break;
    case 14: status = ((int32_t(*)(u64, u64, u64, u64, u64, u64, u64, u64, u64, u64, u64, u64, u64, u64))fn)(a[1], a[2], a[3], a[4], a[5], a[6], a[7], a[8], a[9], a[10], a[11], a[12], a[13], a[14]); break;
    default: status = -1; break;
  }
  Py_END_ALLOW_THREADS
  return PyLong_FromLong(status);
}

/* emb_obs_stack(src, env_ids, n, pixels, channels, layout, out_dtype, scale, offset, dst, stream) */
static PyObject* call_obs_stack(PyObject* self, PyObject* const* args, Py_ssize_t nargs) {
  if (nargs != 12) {
    PyErr_SetString(PyExc_TypeError, "fastcall.obs_stack(addr, 11 arguments)");
    return NULL;
  }
  u64 a[12];
  float f[2];
  for (int i = 0; i < 12; ++i) {
    if (i == 8 || i == 9) {
      if (as_float(args[i], &f[i - 8]) < 0) return NULL;
    } else if (as_u64(args[i], &a[i]) < 0) {
      return NULL;
    }
  }
  void* fn = (void*)(uintptr_t)a[0];
  int32_t status;
  Py_BEGIN_ALLOW_THREADS
  status = ((int32_t(*)(u64, u64, u64, u64, u64, u64, u64, float, float, u64, u64))fn)(
      a[1], a[2], a[3], a[4], a[5], a[6], a[7], f[0], f[1], a[10], a[11]);
  Py_END_ALLOW_THREADS
  return PyLong_FromLong(status);
}

/* emb_scan_gae(rew, val, last, term, B, T, live_scale, lam, adv, tar, stream)      11
 * emb_scan_lambda(last, term, rew, boot, B, T, disc, lam, ret, stream)             10
 * emb_scan_gae_grouped(rew, val, last, term, B, T, live, lam, adv, tar, group, group_stride, stream)  13 */
static PyObject* call_scan(PyObject* self, PyObject* const* args, Py_ssize_t nargs) {
  if (nargs != 11 && nargs != 12 && nargs != 14) {
    PyErr_SetString(PyExc_TypeError, "fastcall.scan(addr, 10, 11 or 13 arguments)");
    return NULL;
  }
  u64 a[14] = {0};
  float f[2];
  for (Py_ssize_t i = 0; i < nargs; ++i) {
    if (i == 7 || i == 8) {
      if (as_float(args[i], &f[i - 7]) < 0) return NULL;
    } else if (as_u64(args[i], &a[i]) < 0) {
      return NULL;
    }
  }
  void* fn = (void*)(uintptr_t)a[0];
  int32_t status;
  Py_BEGIN_ALLOW_THREADS
  if (nargs == 14)   /* emb_scan_gae_grouped: ..., adv, tar, group, group_stride, stream */
    status = ((int32_t(*)(u64, u64, u64, u64, u64, u64, float, float, u64, u64, u64, u64, u64))fn)(
        a[1], a[2], a[3], a[4], a[5], a[6], f[0], f[1], a[9], a[10], a[11], a[12], a[13]);
  else if (nargs == 12)
    status = ((int32_t(*)(u64, u64, u64, u64, u64, u64, float, float, u64, u64, u64))fn)(
        a[1], a[2], a[3], a[4], a[5], a[6], f[0], f[1], a[9], a[10], a[11]);
  else
    status = ((int32_t(*)(u64, u64, u64, u64, u64, u64, float, float, u64, u64))fn)(
        a[1], a[2], a[3], a[4], a[5], a[6], f[0], f[1], a[9], a[10]);
  Py_END_ALLOW_THREADS
  return PyLong_FromLong(status);
}

/* columns(steps: dict, plan: tuple, out: writable buffer of void*, tensor_type,
 *         device_index) -> None | list of positions
 *
 * Replay.add_batch's per-key loop: for every (name, value) of `steps`, in order,
 * with plan[i] = (column, dtype, shape, name[, device]): if `value` is exactly a
 * `tensor_type` of that dtype and shape, contiguous, on GPU `device_index`, its
 * data_ptr() goes to out[column]; otherwise the position is reported back and
 * the caller converts that value the slow way.  column < 0 = not stored.
 * The same checks as the Python loop, minus the interpreter between them.
 *
 * A tensor OBJECT that passed the checks against plan[i] is marked with that
 * plan entry (attribute `_emb_ok`): vector envs hand out the same few tensor
 * objects step after step, and a tensor's dtype, shape, device and strides do
 * not change behind its back (short of resize_ / set_ / `.data =`, which
 * nothing does to an observation), so the next call takes its data_ptr() after
 * one identity comparison instead of four attribute calls. */
static PyObject *s_dtype, *s_shape, *s_is_contiguous, *s_get_device, *s_data_ptr, *s_emb_ok;

/* value.__dict__.get('_emb_ok'), borrowed, without creating the dict or raising. */
static PyObject* ready_mark(PyObject* value) {
  PyObject** dict = _PyObject_GetDictPtr(value);
  if (!dict || !*dict) return NULL;
  return PyDict_GetItemWithError(*dict, s_emb_ok);
}

static PyObject* call_columns(PyObject* self, PyObject* const* args, Py_ssize_t nargs) {
  if (nargs != 5 || !PyDict_Check(args[0]) || !PyTuple_Check(args[1])) {
    PyErr_SetString(PyExc_TypeError, "fastcall.columns(dict, plan tuple, buffer, type, device)");
    return NULL;
  }
  PyObject* steps = args[0];
  PyObject* plan = args[1];
  PyTypeObject* tensor_type = (PyTypeObject*)args[3];
  const long device = PyLong_AsLong(args[4]);
  if (device == -1 && PyErr_Occurred()) return NULL;
  if (PyDict_GET_SIZE(steps) != PyTuple_GET_SIZE(plan)) {
    PyErr_SetString(PyExc_ValueError, "fastcall.columns: plan does not match the dict");
    return NULL;
  }
  Py_buffer view;
  if (PyObject_GetBuffer(args[2], &view, PyBUF_WRITABLE) < 0) return NULL;
  u64* out = (u64*)view.buf;
  const Py_ssize_t slots = view.len / (Py_ssize_t)sizeof(u64);
  PyObject* slow = NULL;
  PyObject *name, *value;
  Py_ssize_t pos = 0, i = 0;
  int failed = 0;
  while (!failed && PyDict_Next(steps, &pos, &name, &value)) {
    PyObject* item = PyTuple_GET_ITEM(plan, i);
    const long column = PyLong_AsLong(PyTuple_GET_ITEM(item, 0));
    if (column < 0 || column >= slots) {
      if (column >= slots) { PyErr_SetString(PyExc_IndexError, "fastcall.columns: column"); failed = 1; }
      ++i;
      continue;
    }
    int ok = Py_TYPE(value) == tensor_type;
    PyObject* r;
    /* Marks are taken and left only for plan entries that name their device
     * (a fifth element equal to `device`). */
    const int markable = PyTuple_GET_SIZE(item) >= 5 &&
                         PyLong_AsLong(PyTuple_GET_ITEM(item, 4)) == device;
    const int marked = ok && markable && ready_mark(value) == item;
    if (ok && !marked && PyErr_Occurred()) { failed = 1; break; }
    if (ok && !marked) {
      r = PyObject_GetAttr(value, s_dtype);
      if (!r) { failed = 1; break; }
      ok = r == PyTuple_GET_ITEM(item, 1);
      Py_DECREF(r);
    }
    if (ok && !marked) {
      r = PyObject_GetAttr(value, s_shape);
      if (!r) { failed = 1; break; }
      const int eq = PyObject_RichCompareBool(r, PyTuple_GET_ITEM(item, 2), Py_EQ);
      Py_DECREF(r);
      if (eq < 0) { failed = 1; break; }
      ok = eq;
    }
    if (ok && !marked) {
      r = PyObject_CallMethodNoArgs(value, s_is_contiguous);
      if (!r) { failed = 1; break; }
      ok = r == Py_True;
      Py_DECREF(r);
    }
    if (ok && !marked) {
      r = PyObject_CallMethodNoArgs(value, s_get_device);
      if (!r) { failed = 1; break; }
      const long where = PyLong_AsLong(r);
      Py_DECREF(r);
      if (where == -1 && PyErr_Occurred()) { failed = 1; break; }
      ok = where == device;
      if (ok && markable && PyObject_SetAttr(value, s_emb_ok, item) < 0)
        PyErr_Clear();                               /* no __dict__: stay unmarked */
    }
    if (ok) {
      r = PyObject_CallMethodNoArgs(value, s_data_ptr);
      if (!r) { failed = 1; break; }
      u64 address = 0;
      const int bad = as_u64(r, &address);
      Py_DECREF(r);
      if (bad < 0) { failed = 1; break; }
      out[column] = address;
    } else {
      if (!slow && !(slow = PyList_New(0))) { failed = 1; break; }
      PyObject* index = PyLong_FromSsize_t(i);
      if (!index || PyList_Append(slow, index) < 0) { Py_XDECREF(index); failed = 1; break; }
      Py_DECREF(index);
    }
    ++i;
  }
  PyBuffer_Release(&view);
  if (failed) {
    Py_XDECREF(slow);
    return NULL;
  }
  if (slow) return slow;
  Py_RETURN_NONE;
}

/* same_values(d: dict, t: tuple) -> bool: len(d) == len(t) and the i-th value of
 * d IS t[i] for every i.  A vector env hands out the same few tensor objects
 * step after step: one call tells whether this step's observations are the
 * objects a cached step record was made for. */
static PyObject* call_same_values(PyObject* self, PyObject* const* args, Py_ssize_t nargs) {
  if (nargs != 2 || !PyDict_Check(args[0]) || !PyTuple_Check(args[1])) {
    PyErr_SetString(PyExc_TypeError, "fastcall.same_values(dict, tuple)");
    return NULL;
  }
  if (PyDict_GET_SIZE(args[0]) != PyTuple_GET_SIZE(args[1])) Py_RETURN_FALSE;
  PyObject *name, *value;
  Py_ssize_t pos = 0, i = 0;
  while (PyDict_Next(args[0], &pos, &name, &value))
    if (value != PyTuple_GET_ITEM(args[1], i++)) Py_RETURN_FALSE;
  Py_RETURN_TRUE;
}

/* ---- Replay.add of one host step dict (replay.py:77-118), staged in C --------
 *
 * stage_plan(keys, sid_base, sid_bytes, dst) -> capsule.  `keys` lists the
 * schema's columns but the step id: (name, stage base address, row bytes,
 * kind, item size, shape) with kind one of 'b' bool, 'i' signed, 'u' unsigned,
 * 'f' float.  `sid_base` / `dst`: the pinned step-id rows and the int32 pool
 * rows of the staged steps.
 *
 * add_step(plan, step, slot, fn, handle, workers, rows, sids, new_chunks[, n]) -> int
 *   checks every value of the dict against the plan (same key set but `log/*`,
 *   buffer protocol, C-contiguous, same kind / item size / shape) BEFORE the
 *   index is touched, calls emb_replay_add_index(handle, 1, workers, rows, sids,
 *   new_chunks) by address and, on status 0, copies the values, the step id and
 *   the pool row into stage row `slot`.  Returns 0, the library's status (> 0:
 *   nothing was staged), or -1 when the step needs the Python path (a value
 *   without a buffer, another dtype, a shape or key mismatch -- which that path
 *   converts or reports exactly as before); nothing has been touched then.
 *   With n > 0 the dict holds n steps, one per workers[i] (Replay.add_batch of
 *   host arrays): every value has a leading dimension n, the index is called
 *   for n steps and stage rows [slot, slot + n) are filled. */
#define STAGE_MAX_KEYS 64
#define STAGE_MAX_DIMS 8

typedef struct {
  PyObject* name;
  char* base;
  int64_t rowbytes;
  int itemsize, ndim;
  char kind;
  Py_ssize_t shape[STAGE_MAX_DIMS];
} StageKey;

typedef struct {
  int n;
  StageKey keys[STAGE_MAX_KEYS];
  char* sid_base;
  int64_t sid_bytes;
  int32_t* dst;
} StagePlan;

static void stage_plan_free(PyObject* capsule) {
  StagePlan* plan = (StagePlan*)PyCapsule_GetPointer(capsule, "emb.stage_plan");
  if (!plan) return;
  for (int i = 0; i < plan->n; ++i) Py_XDECREF(plan->keys[i].name);
  PyMem_Free(plan);
}

static PyObject* call_stage_plan(PyObject* self, PyObject* const* args, Py_ssize_t nargs) {
  if (nargs != 4 || !PyTuple_Check(args[0])) {
    PyErr_SetString(PyExc_TypeError, "fastcall.stage_plan(keys tuple, sid_base, sid_bytes, dst)");
    return NULL;
  }
  const Py_ssize_t n = PyTuple_GET_SIZE(args[0]);
  if (n < 1 || n > STAGE_MAX_KEYS) {
    PyErr_SetString(PyExc_ValueError, "fastcall.stage_plan: 1 .. 64 keys");
    return NULL;
  }
  u64 sid_base, sid_bytes, dst;
  if (as_u64(args[1], &sid_base) < 0 || as_u64(args[2], &sid_bytes) < 0 || as_u64(args[3], &dst) < 0)
    return NULL;
  StagePlan* plan = (StagePlan*)PyMem_Calloc(1, sizeof(StagePlan));
  if (!plan) return PyErr_NoMemory();
  plan->sid_base = (char*)(uintptr_t)sid_base;
  plan->sid_bytes = (int64_t)sid_bytes;
  plan->dst = (int32_t*)(uintptr_t)dst;
  for (Py_ssize_t i = 0; i < n; ++i) {
    PyObject* item = PyTuple_GET_ITEM(args[0], i);
    const char* kind;
    PyObject *name, *shape;
    unsigned long long base;
    long long rowbytes;
    int itemsize;
    if (!PyTuple_Check(item) ||
        !PyArg_ParseTuple(item, "UKLsiO!", &name, &base, &rowbytes, &kind, &itemsize, &PyTuple_Type, &shape) ||
        PyTuple_GET_SIZE(shape) > STAGE_MAX_DIMS || !kind[0]) {
      if (!PyErr_Occurred()) PyErr_SetString(PyExc_ValueError, "fastcall.stage_plan: bad key entry");
      goto fail;
    }
    StageKey* key = &plan->keys[i];
    key->base = (char*)(uintptr_t)base;
    key->rowbytes = rowbytes;
    key->itemsize = itemsize;
    key->kind = kind[0];
    key->ndim = (int)PyTuple_GET_SIZE(shape);
    int64_t count = 1;
    for (int d = 0; d < key->ndim; ++d) {
      key->shape[d] = PyLong_AsSsize_t(PyTuple_GET_ITEM(shape, d));
      if (key->shape[d] < 0) {
        if (!PyErr_Occurred()) PyErr_SetString(PyExc_ValueError, "fastcall.stage_plan: bad shape");
        goto fail;
      }
      count *= key->shape[d];
    }
    if (count * itemsize != rowbytes) {
      PyErr_SetString(PyExc_ValueError, "fastcall.stage_plan: row bytes != items x item size");
      goto fail;
    }
    Py_INCREF(name);
    key->name = name;
    plan->n = (int)i + 1;
  }
  {
    PyObject* capsule = PyCapsule_New(plan, "emb.stage_plan", stage_plan_free);
    if (capsule) return capsule;
  }
fail:
  for (int i = 0; i < plan->n; ++i) Py_XDECREF(plan->keys[i].name);
  PyMem_Free(plan);
  return NULL;
}

/* struct-module format of a buffer -> kind; 0: not a native fixed-size scalar. */
static char format_kind(const char* format) {
  if (!format) return 'u';                          /* PyBUF_SIMPLE exporters: bytes */
  if (*format == '@' || *format == '=' || *format == '<' || *format == '|') ++format;
  if (!format[0] || format[1]) return 0;
  switch (format[0]) {
    case '?': return 'b';
    case 'b': case 'h': case 'i': case 'l': case 'q': case 'n': return 'i';
    case 'B': case 'H': case 'I': case 'L': case 'Q': case 'N': return 'u';
    case 'e': case 'f': case 'd': return 'f';
    default: return 0;
  }
}

static PyObject* call_add_step(PyObject* self, PyObject* const* args, Py_ssize_t nargs) {
  if (nargs != 9 && nargs != 10) {
    PyErr_SetString(PyExc_TypeError,
                    "fastcall.add_step(plan, step, slot, fn, handle, workers, rows, sids, new_chunks[, n])");
    return NULL;
  }
  u64 batch = 0;
  if (nargs == 10 && as_u64(args[9], &batch) < 0) return NULL;
  const int lead = batch > 0;                       /* values carry a leading dimension n */
  const int64_t steps = lead ? (int64_t)batch : 1;
  StagePlan* plan = (StagePlan*)PyCapsule_GetPointer(args[0], "emb.stage_plan");
  if (!plan) return NULL;
  PyObject* step = args[1];
  if (!PyDict_Check(step)) return PyLong_FromLong(-1);
  u64 a[7];
  for (int i = 0; i < 7; ++i)
    if (as_u64(args[2 + i], &a[i]) < 0) return NULL;
  const int64_t slot = (int64_t)a[0];
  Py_buffer views[STAGE_MAX_KEYS];
  int column[STAGE_MAX_KEYS];
  int held = 0, slow = 0;
  u64 seen = 0;
  PyObject *name, *value;
  Py_ssize_t pos = 0;
  int guess = 0;                                    /* dicts keep the schema's order as a rule */
  while (PyDict_Next(step, &pos, &name, &value)) {
    if (!PyUnicode_Check(name)) { slow = 1; break; }
    int found = -1;
    if (guess < plan->n && plan->keys[guess].name == name) {
      found = guess;
    } else {
      for (int j = 0; j < plan->n && found < 0; ++j)
        if (plan->keys[j].name == name) found = j;
      for (int j = 0; j < plan->n && found < 0; ++j)
        if (PyUnicode_Compare(plan->keys[j].name, name) == 0) found = j;
      if (found < 0) {
        if (PyErr_Occurred()) PyErr_Clear();
        const Py_ssize_t len = PyUnicode_GET_LENGTH(name);
        if (len >= 4 && PyUnicode_READ_CHAR(name, 0) == 'l' && PyUnicode_READ_CHAR(name, 1) == 'o' &&
            PyUnicode_READ_CHAR(name, 2) == 'g' && PyUnicode_READ_CHAR(name, 3) == '/')
          continue;                                 /* log/* keys are not stored (replay.py:78) */
        slow = 1;                                   /* unknown key: the Python path raises */
        break;
      }
    }
    guess = found + 1;
    if (seen >> found & 1) { slow = 1; break; }
    seen |= (u64)1 << found;
    const StageKey* key = &plan->keys[found];
    if (!PyObject_CheckBuffer(value)) { slow = 1; break; }
    Py_buffer* view = &views[held];
    if (PyObject_GetBuffer(value, view, PyBUF_FORMAT | PyBUF_C_CONTIGUOUS) < 0) {
      PyErr_Clear();
      slow = 1;
      break;
    }
    column[held++] = found;
    if (view->ndim != key->ndim + lead || view->itemsize != key->itemsize ||
        view->len != steps * key->rowbytes || format_kind(view->format) != key->kind) { slow = 1; break; }
    if (lead && view->shape[0] != steps) { slow = 1; break; }
    for (int d = 0; d < key->ndim; ++d)
      if (view->shape[d + lead] != key->shape[d]) slow = 1;
    if (slow) break;
  }
  if (!slow && held != plan->n) slow = 1;           /* a key is missing: the Python path raises */
  int32_t status = -1;
  if (!slow) {
    void* fn = (void*)(uintptr_t)a[1];
    Py_BEGIN_ALLOW_THREADS
    status = ((int32_t(*)(u64, u64, u64, u64, u64, u64))fn)(a[2], (u64)steps, a[3], a[4], a[5], a[6]);
    if (status == 0) {
      for (int i = 0; i < held; ++i) {
        const StageKey* key = &plan->keys[column[i]];
        memcpy(key->base + slot * key->rowbytes, views[i].buf, (size_t)(steps * key->rowbytes));
      }
      memcpy(plan->sid_base + slot * plan->sid_bytes, (const void*)(uintptr_t)a[5],
             (size_t)(steps * plan->sid_bytes));
      memcpy(plan->dst + slot, (const void*)(uintptr_t)a[4], (size_t)steps * sizeof(int32_t));
    }
    Py_END_ALLOW_THREADS
  }
  for (int i = 0; i < held; ++i) PyBuffer_Release(&views[i]);
  return PyLong_FromLong(status);
}

static PyMethodDef methods[] = {
    {"ints", (PyCFunction)(void (*)(void))call_ints, METH_FASTCALL,
     "ints(addr, *args) -> status: call an int32 f(pointers/integers...)"},
    {"obs_stack", (PyCFunction)(void (*)(void))call_obs_stack, METH_FASTCALL,
     "obs_stack(addr, *11 args) -> status"},
    {"scan", (PyCFunction)(void (*)(void))call_scan, METH_FASTCALL,
     "scan(addr, *10 or 11 args) -> status"},
    {"columns", (PyCFunction)(void (*)(void))call_columns, METH_FASTCALL,
     "columns(steps, plan, out, tensor_type, device) -> None | positions for the slow path"},
    {"stage_plan", (PyCFunction)(void (*)(void))call_stage_plan, METH_FASTCALL,
     "stage_plan(keys, sid_base, sid_bytes, dst) -> capsule for add_step"},
    {"add_step", (PyCFunction)(void (*)(void))call_add_step, METH_FASTCALL,
     "add_step(plan, step, slot, fn, handle, workers, rows, sids, new_chunks) -> 0 | status | -1 (Python path)"},
    {"same_values", (PyCFunction)(void (*)(void))call_same_values, METH_FASTCALL,
     "same_values(dict, tuple) -> the dict's values are exactly these objects, in order"},
    {NULL, NULL, 0, NULL},
};

static struct PyModuleDef module = {
    PyModuleDef_HEAD_INIT, "_emb_fastcall",
    "Low-overhead calls into libembodied_hip.so by function address.", -1, methods,
};

PyMODINIT_FUNC PyInit__emb_fastcall(void) {
  s_dtype = PyUnicode_InternFromString("dtype");
  s_shape = PyUnicode_InternFromString("shape");
  s_is_contiguous = PyUnicode_InternFromString("is_contiguous");
  s_get_device = PyUnicode_InternFromString("get_device");
  s_data_ptr = PyUnicode_InternFromString("data_ptr");
  s_emb_ok = PyUnicode_InternFromString("_emb_ok");
  if (!s_dtype || !s_shape || !s_is_contiguous || !s_get_device || !s_data_ptr || !s_emb_ok) return NULL;
  return PyModule_Create(&module);
}
