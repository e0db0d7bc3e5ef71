// CPython extension: the per-parameter "gradient ready" hook of DistributedDataParallel as a C callable.
//
// Autograd calls one hook per parameter per backward (161 for a ResNet-50, 62 for a ResNet-18).  As a Python
// closure that reads `param.grad`, checks it against the bucket view, and goes through ctypes into the C++ bucket
// plan, a call costs 5-6 us of host time - 0.9 ms per step, a third of the host time of a launch-bound CIFAR step.
// Here the call is one C function: the identity check of `param.grad`, the plan's mark_ready() through a plain
// function pointer, and Python only when something has to happen (the first gradient of a backward queues the
// finalize callback, a completed bucket launches its all-reduce, a replaced `.grad` is folded back).
// (Reference behaviour: torch's C++ Reducer::autograd_hook under /root/reference/2.distributed.py:107.)
#define PY_SSIZE_T_CLEAN
#include <Python.h>
#include <structmember.h>

namespace {

typedef int (*mark_ready_fn)(void* plan, int param, int* out, int cap);

struct HookState {
  PyObject_HEAD
  int enabled;   // gradient synchronisation is on (not inside no_sync(), communicator active)
  int queued;    // the finalize callback of this backward has been queued
};

PyMemberDef HookState_members[] = {
    {"enabled", T_INT, offsetof(HookState, enabled), 0, "hooks do anything at all"},
    {"queued", T_INT, offsetof(HookState, queued), 0, "finalize callback queued for this backward"},
    {nullptr, 0, 0, 0, nullptr}};

PyObject* HookState_new(PyTypeObject* type, PyObject*, PyObject*) {
  HookState* s = reinterpret_cast<HookState*>(type->tp_alloc(type, 0));
  if (s != nullptr) { s->enabled = 1; s->queued = 0; }
  return reinterpret_cast<PyObject*>(s);
}

PyTypeObject HookStateType = {PyVarObject_HEAD_INIT(nullptr, 0)};

struct ReadyHook {
  PyObject_HEAD
  HookState* state;
  void* plan;
  mark_ready_fn mark;
  int index;
  int cap;
  int* out;
  PyObject* view;      // the bucket view that param.grad is expected to be (or None: no check)
  PyObject* on_first;  // callable(): first ready gradient of a backward
  PyObject* launch;    // callable(bucket_index)
  PyObject* fold;      // callable(index, param): param.grad is not the bucket view
};

PyObject* g_str_grad = nullptr;

int ReadyHook_traverse(ReadyHook* h, visitproc visit, void* arg) {
  Py_VISIT(reinterpret_cast<PyObject*>(h->state));
  Py_VISIT(h->view);
  Py_VISIT(h->on_first);
  Py_VISIT(h->launch);
  Py_VISIT(h->fold);
  return 0;
}

int ReadyHook_clear(ReadyHook* h) {
  Py_CLEAR(h->state);
  Py_CLEAR(h->view);
  Py_CLEAR(h->on_first);
  Py_CLEAR(h->launch);
  Py_CLEAR(h->fold);
  return 0;
}

void ReadyHook_dealloc(ReadyHook* h) {
  PyObject_GC_UnTrack(h);
  ReadyHook_clear(h);
  PyMem_Free(h->out);
  Py_TYPE(h)->tp_free(reinterpret_cast<PyObject*>(h));
}

// ReadyHook(state, plan_handle, mark_ready_address, index, n_params, view, on_first, launch, fold)
PyObject* ReadyHook_new(PyTypeObject* type, PyObject* args, PyObject*) {
  PyObject *state, *view, *on_first, *launch, *fold;
  unsigned long long plan, mark;
  int index, cap;
  if (!PyArg_ParseTuple(args, "O!KKiiOOOO", &HookStateType, &state, &plan, &mark, &index, &cap, &view, &on_first,
                        &launch, &fold))
    return nullptr;
  if (plan == 0 || mark == 0 || cap < 1) {
    PyErr_SetString(PyExc_ValueError, "ReadyHook: null plan / mark_ready address or empty plan");
    return nullptr;
  }
  ReadyHook* h = reinterpret_cast<ReadyHook*>(type->tp_alloc(type, 0));
  if (h == nullptr) return nullptr;
  h->out = static_cast<int*>(PyMem_Malloc(sizeof(int) * static_cast<size_t>(cap)));
  if (h->out == nullptr) { Py_DECREF(h); return PyErr_NoMemory(); }
  Py_INCREF(state); h->state = reinterpret_cast<HookState*>(state);
  h->plan = reinterpret_cast<void*>(static_cast<uintptr_t>(plan));
  h->mark = reinterpret_cast<mark_ready_fn>(static_cast<uintptr_t>(mark));
  h->index = index;
  h->cap = cap;
  Py_INCREF(view); h->view = view;
  Py_INCREF(on_first); h->on_first = on_first;
  Py_INCREF(launch); h->launch = launch;
  Py_INCREF(fold); h->fold = fold;
  return reinterpret_cast<PyObject*>(h);
}

// hook(param) from autograd's post-accumulate-grad hook, hook() from an op that wrote the gradient in place
PyObject* ReadyHook_call(ReadyHook* h, PyObject* args, PyObject*) {
  if (!h->state->enabled) Py_RETURN_NONE;
  if (PyTuple_GET_SIZE(args) >= 1 && h->view != Py_None) {
    PyObject* param = PyTuple_GET_ITEM(args, 0);
    PyObject* g = PyObject_GetAttr(param, g_str_grad);
    if (g == nullptr) return nullptr;
    const bool same = g == h->view || g == Py_None;
    Py_DECREF(g);
    if (!same) {   // someone replaced .grad: Python folds it back into the bucket view
      PyObject* idx = PyLong_FromLong(h->index);
      if (idx == nullptr) return nullptr;
      PyObject* r = PyObject_CallFunctionObjArgs(h->fold, idx, param, nullptr);
      Py_DECREF(idx);
      if (r == nullptr) return nullptr;
      Py_DECREF(r);
    }
  }
  if (!h->state->queued) {
    h->state->queued = 1;
    PyObject* r = PyObject_CallNoArgs(h->on_first);
    if (r == nullptr) return nullptr;
    Py_DECREF(r);
  }
  const int k = h->mark(h->plan, h->index, h->out, h->cap);
  for (int i = 0; i < k; ++i) {
    PyObject* b = PyLong_FromLong(h->out[i]);
    if (b == nullptr) return nullptr;
    PyObject* r = PyObject_CallOneArg(h->launch, b);
    Py_DECREF(b);
    if (r == nullptr) return nullptr;
    Py_DECREF(r);
  }
  Py_RETURN_NONE;
}

PyMemberDef ReadyHook_members[] = {
    {"view", T_OBJECT_EX, offsetof(ReadyHook, view), 0, "bucket view param.grad is expected to be (None: unchecked)"},
    {"index", T_INT, offsetof(ReadyHook, index), READONLY, "parameter index in the plan"},
    {nullptr, 0, 0, 0, nullptr}};

PyTypeObject ReadyHookType = {PyVarObject_HEAD_INIT(nullptr, 0)};

PyModuleDef module_def = {PyModuleDef_HEAD_INIT, "_pda_pyext", "C callables of pytorch_distributed_amd", -1, nullptr,
                          nullptr, nullptr, nullptr, nullptr};

}  // namespace

PyMODINIT_FUNC PyInit__pda_pyext(void) {
  HookStateType.tp_name = "_pda_pyext.HookState";
  HookStateType.tp_basicsize = sizeof(HookState);
  HookStateType.tp_flags = Py_TPFLAGS_DEFAULT;
  HookStateType.tp_new = HookState_new;
  HookStateType.tp_members = HookState_members;
  ReadyHookType.tp_name = "_pda_pyext.ReadyHook";
  ReadyHookType.tp_basicsize = sizeof(ReadyHook);
  ReadyHookType.tp_flags = Py_TPFLAGS_DEFAULT | Py_TPFLAGS_HAVE_GC;
  ReadyHookType.tp_new = ReadyHook_new;
  ReadyHookType.tp_dealloc = reinterpret_cast<destructor>(ReadyHook_dealloc);
  ReadyHookType.tp_traverse = reinterpret_cast<traverseproc>(ReadyHook_traverse);
  ReadyHookType.tp_clear = reinterpret_cast<inquiry>(ReadyHook_clear);
  ReadyHookType.tp_call = reinterpret_cast<ternaryfunc>(ReadyHook_call);
  ReadyHookType.tp_members = ReadyHook_members;
  if (PyType_Ready(&HookStateType) < 0 || PyType_Ready(&ReadyHookType) < 0) return nullptr;
  PyObject* m = PyModule_Create(&module_def);
  if (m == nullptr) return nullptr;
  g_str_grad = PyUnicode_InternFromString("grad");
  Py_INCREF(&HookStateType);
  Py_INCREF(&ReadyHookType);
  if (g_str_grad == nullptr || PyModule_AddObject(m, "HookState", reinterpret_cast<PyObject*>(&HookStateType)) < 0 ||
      PyModule_AddObject(m, "ReadyHook", reinterpret_cast<PyObject*>(&ReadyHookType)) < 0) {
    Py_DECREF(m);
    return nullptr;
  }
  return m;
}
