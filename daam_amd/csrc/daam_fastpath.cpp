// daam_amd._fastpath -- host-side recorder of deferred taps (CPython extension, no pybind).
//
// The attention processor calls HeatMapEngine.tap_qk once per cross-attention layer per denoising
// step: 3000 calls per 50-step SDXL generation (reference: the body of
// UNetCrossAttentionHooker.__call__, daam/trace.py:285-294).  In deferred mode a call only has to
// (1) check that the layer's Q / K look like they did when the layer's DaamQKDesc was built and
// (2) remember two device pointers and keep their memory alive until the launch.  Done in Python that
// is ~0.6 us per call, as much as the GPU needs for the tap itself; here it is one METH_FASTCALL
// function that reads the at::Tensor fields directly.  What is kept is the c10::Storage, not the tensor:
// the Python wrapper and the TensorImpl die where they would have died without the trace, only the
// allocator block outlives the processor call.  After the launch the 6000 storages of an SDXL generation
// go to a reaper thread (no GIL needed to return a block to the caching allocator), because freeing
// them inline cost 4.8 ms of host time per generation -- twice the GPU time of the tap itself.  Everything that is not the steady state
// (first call of a layer, shape / dtype change, non-contiguous input, window full) is handed back to
// the Python engine through callbacks, so the validation logic and error messages live in one place
// (daam_amd/engine.py).  No device work and no libdaam_hip calls happen here: the engine passes the
// recorded arrays to daam_tap_qk_enqueue_many (include/daam_hip.h) at flush time.
#include <Python.h>
#include <torch/csrc/autograd/python_variable.h>
#include <ATen/core/Tensor.h>

#include <ATen/Functions.h>
#include <c10/core/GradMode.h>
#include <c10/core/Storage.h>
#include <c10/hip/HIPStream.h>

#include <pthread.h>

#include <atomic>
#include <climits>
#include <condition_variable>
#include <cstdint>
#include <deque>
#include <exception>
#include <mutex>
#include <thread>
#include <vector>

namespace {

// ---- reaper: returns recorded Q / K storages to the allocator off the interpreter's thread -------------------
// One batch = the storages of one launch.  At most one batch waits while another is being freed: drop() blocks
// (GIL released) until the previous batch is gone, so the memory that outlives a launch stays bounded by one
// launch's worth.  A StorageImpl that owns a Python object (someone asked for tensor.untyped_storage()) takes
// the GIL inside its own destructor; that is torch's business and merely slower.
class Reaper {
public:
    using Batch = std::vector<c10::Storage>;

    void submit(Batch&& batch) {
        if (batch.empty()) return;
        if (!usable()) { batch.clear(); return; }                 // synchronous: after shutdown(), in a forked child
        std::unique_lock<std::mutex> lock(mu_);
        if (!started_) {
            try {
                pid_thread_ = std::thread([this] { run(); });
                started_ = true;
            } catch (...) {                                       // no thread to be had: free inline from now on
                off_.store(true);
                lock.unlock();
                batch.clear();
                return;
            }
        }
        idle_.wait(lock, [this] { return queue_.empty(); });
        queue_.push_back(std::move(batch));
        busy_ = true;
        work_.notify_one();
    }

    void drain() {                                                // every submitted batch has been freed
        if (!usable()) return;
        std::unique_lock<std::mutex> lock(mu_);
        idle_.wait(lock, [this] { return queue_.empty() && !busy_; });
    }

    void shutdown() {                                             // atexit: no thread may touch the allocator later
        if (!usable()) return;
        {
            std::unique_lock<std::mutex> lock(mu_);
            idle_.wait(lock, [this] { return queue_.empty() && !busy_; });
            stop_ = true;
            work_.notify_one();
        }
        if (started_ && pid_thread_.joinable()) pid_thread_.join();
        off_.store(true);
    }

    void disable_after_fork() { off_.store(true); }               // the thread does not exist in the child
    void set_sync(bool sync) { sync_.store(sync); }
    bool usable() const { return !off_.load() && !sync_.load(); }

private:
    void run() {
        std::unique_lock<std::mutex> lock(mu_);
        for (;;) {
            work_.wait(lock, [this] { return stop_ || !queue_.empty(); });
            if (queue_.empty()) return;                           // stop_ with nothing queued
            Batch batch = std::move(queue_.front());
            queue_.pop_front();
            idle_.notify_all();                                   // the next batch may queue up behind this one
            lock.unlock();
            batch.clear();                                        // ~StorageImpl -> allocator free, no GIL held
            batch.shrink_to_fit();
            lock.lock();
            if (queue_.empty()) busy_ = false;
            idle_.notify_all();
        }
    }

    std::mutex mu_;
    std::condition_variable work_, idle_;
    std::deque<Batch> queue_;
    std::thread pid_thread_;
    bool started_ = false, stop_ = false, busy_ = false;
    std::atomic<bool> off_{false}, sync_{false};
};

Reaper& reaper() {
    static Reaper* r = new Reaper();                              // never destroyed: no static-destruction-order races
    return *r;
}

// A layer's last validated call (tap or attend): shapes, dtype, device, parameters and the address of the descriptor the
// engine built for it -- the C++ twin of daam_amd/engine.py::CallShape (set_cache / set_attend_cache copy one across).
struct CallShape {
    bool valid = false;
    int64_t q_size[3] = {0, 0, 0};
    int64_t k_size[3] = {0, 0, 0};
    int dtype = -1;              // c10::ScalarType
    int device_type = -1;        // c10::DeviceType of the validated call (the engine only admits HIP devices)
    int device = -1;
    long heads = 0, factor = 0;  // factor: taps only
    int round_logits = 1;
    double scale = 0.0;
    uint64_t desc = 0;           // address of the DaamQKDesc / DaamAttendDesc (owned by the engine's CallShape)
};
using LayerCache = CallShape;
using AttendCache = CallShape;

// daam_attend (include/daam_hip.h), reached through the address the engine takes from the loaded library
using AttendFn = int (*)(void* ctx, int layer, const void* q, const void* k, const void* v, void* out, const void* desc, int tap,
                         void* stream);

struct Recorder {
    PyObject_HEAD
    std::vector<LayerCache>* cache;
    std::vector<AttendCache>* acache;
    void* native_ctx;                   // DaamCtx* of the engine (0: none yet)
    AttendFn attend_fn;                 // daam_attend, or null: every attend() goes to Python
    PyObject* attend_slow_cb;           // engine._attend_slow(layer, q, k, v, heads, scale, factor, round_logits, tapped)
    std::vector<int32_t>* cnt;          // recorded steps per layer
    std::vector<uint8_t>* touched;      // engine._touch(layer) already called since clear()
    std::vector<int32_t>* layers;       // parallel arrays handed to daam_tap_qk_enqueue_many
    std::vector<uint64_t>* qp;
    std::vector<uint64_t>* kp;
    std::vector<uint64_t>* dp;
    std::vector<c10::Storage>* keep;    // memory of the recorded Q / K stays allocated until drop()
    int window;                         // steps of one layer per launch
    int64_t budget;                     // bytes of recorded Q / K kept alive before a launch is forced
    int64_t held;                       // bytes of Q / K currently recorded
    PyObject* slow_cb;                  // engine._tap_qk_slow(layer, q, k, heads, scale, factor, round_logits)
    PyObject* flush_cb;                 // engine.flush()
    PyObject* touch_cb;                 // engine._touch(layer)
};

int Recorder_init(Recorder* self, PyObject* args, PyObject* kw) {
    static const char* names[] = {"n_layers", "slow", "flush", "touch", nullptr};
    int n = 0;
    PyObject *slow, *flush, *touch;
    if (!PyArg_ParseTupleAndKeywords(args, kw, "iOOO", const_cast<char**>(names), &n, &slow, &flush, &touch)) return -1;
    if (n < 0) { PyErr_SetString(PyExc_ValueError, "n_layers < 0"); return -1; }
    self->cache = new std::vector<LayerCache>(n);
    self->acache = new std::vector<AttendCache>(n);
    self->native_ctx = nullptr;
    self->attend_fn = nullptr;
    self->attend_slow_cb = nullptr;
    self->cnt = new std::vector<int32_t>(n, 0);
    self->touched = new std::vector<uint8_t>(n, 0);
    self->layers = new std::vector<int32_t>();
    self->qp = new std::vector<uint64_t>();
    self->kp = new std::vector<uint64_t>();
    self->dp = new std::vector<uint64_t>();
    self->keep = new std::vector<c10::Storage>();
    self->window = 1;
    self->budget = INT64_MAX;
    self->held = 0;
    Py_INCREF(slow); Py_INCREF(flush); Py_INCREF(touch);
    self->slow_cb = slow; self->flush_cb = flush; self->touch_cb = touch;
    return 0;
}

int Recorder_traverse(Recorder* self, visitproc visit, void* arg) {
    Py_VISIT(self->slow_cb); Py_VISIT(self->flush_cb); Py_VISIT(self->touch_cb); Py_VISIT(self->attend_slow_cb);
    return 0;
}

int Recorder_clear(Recorder* self) {
    Py_CLEAR(self->slow_cb); Py_CLEAR(self->flush_cb); Py_CLEAR(self->touch_cb); Py_CLEAR(self->attend_slow_cb);
    return 0;
}

void Recorder_dealloc(Recorder* self) {
    PyObject_GC_UnTrack(self);
    delete self->cache; delete self->acache; delete self->cnt; delete self->touched; delete self->layers;
    delete self->qp; delete self->kp; delete self->dp; delete self->keep;
    Py_XDECREF(self->slow_cb); Py_XDECREF(self->flush_cb); Py_XDECREF(self->touch_cb); Py_XDECREF(self->attend_slow_cb);
    Py_TYPE(self)->tp_free(reinterpret_cast<PyObject*>(self));
}

inline bool same3(const at::Tensor& t, const int64_t (&s)[3]) {
    if (t.dim() != 3) return false;
    const auto sz = t.sizes();
    return sz[0] == s[0] && sz[1] == s[1] && sz[2] == s[2];
}

inline bool same_device(const at::Tensor& t, const LayerCache& c) {
    const auto d = t.device();
    return static_cast<int>(d.type()) == c.device_type && static_cast<int>(d.index()) == c.device;
}

// A launch is due before recording a tap of `layer` when the layer's step window is full, or -- checked
// on step boundaries only (the layer that opened this window comes round again) -- when the recorded
// Q / K have reached the byte budget.
inline bool must_launch(const Recorder* self, long layer) {
    if ((*self->cnt)[layer] >= self->window) return true;
    return self->held >= self->budget && !self->layers->empty() && (*self->layers)[0] == layer;
}

inline void push(Recorder* self, int layer, const at::Tensor& q, const at::Tensor& k, uint64_t desc) {
    self->layers->push_back(layer);
    self->qp->push_back(reinterpret_cast<uint64_t>(q.data_ptr()));
    self->kp->push_back(reinterpret_cast<uint64_t>(k.data_ptr()));
    self->dp->push_back(desc);
    self->keep->push_back(q.storage());
    self->keep->push_back(k.storage());
    self->held += static_cast<int64_t>(q.nbytes()) + static_cast<int64_t>(k.nbytes());
    (*self->cnt)[layer] += 1;
}

// The steady state of a deferred tap: the call looks like the one the layer's descriptor was built for -> record it.
// Returns 1: recorded; 0: not the steady state (the caller hands the call to Python); -1: a Python error is set.
int record_steady(Recorder* self, long layer, PyObject* layer_obj, PyObject* qo, PyObject* ko, long heads, double scale, long factor,
                  int rl) {
    // torch accessors can throw (tensors without storage: fake / meta tensors under tracing); anything
    // unusual goes to the Python path, which reports it properly
    bool steady = false;
    try {
        const LayerCache& c = (*self->cache)[layer];
        const at::Tensor& q = THPVariable_Unpack(qo);
        const at::Tensor& k = THPVariable_Unpack(ko);
        steady = c.valid && same3(q, c.q_size) && same3(k, c.k_size) &&
                 static_cast<int>(q.scalar_type()) == c.dtype && static_cast<int>(k.scalar_type()) == c.dtype &&
                 same_device(q, c) && same_device(k, c) &&
                 heads == c.heads && factor == c.factor && scale == c.scale && rl == c.round_logits &&
                 q.is_contiguous() && k.is_contiguous() && q.has_storage() && k.has_storage();
    } catch (...) {
        steady = false;
    }
    if (!steady) return 0;
    if (!self->flush_cb || !self->touch_cb) { PyErr_SetString(PyExc_RuntimeError, "recorder is closed"); return -1; }
    if (must_launch(self, layer)) {
        PyObject* r = PyObject_CallNoArgs(self->flush_cb);     // launches, then calls drop()
        if (!r) return -1;
        Py_DECREF(r);
    }
    try {
        push(self, static_cast<int>(layer), THPVariable_Unpack(qo), THPVariable_Unpack(ko), (*self->cache)[layer].desc);
    } catch (const std::exception& e) {
        PyErr_SetString(PyExc_RuntimeError, e.what());
        return -1;
    }
    if (!(*self->touched)[layer]) {
        PyObject* r = PyObject_CallOneArg(self->touch_cb, layer_obj);
        if (!r) return -1;
        Py_DECREF(r);
        (*self->touched)[layer] = 1;
    }
    return 1;
}

// tap(layer, query, key, heads, scale, factor, round_logits=True) -> None
PyObject* Recorder_tap(Recorder* self, PyObject* const* args, Py_ssize_t nargs, PyObject* kwnames) {
    const bool has_kw = kwnames != nullptr && PyTuple_GET_SIZE(kwnames) > 0;
    if (!has_kw && (nargs == 6 || nargs == 7) && THPVariable_Check(args[1]) && THPVariable_Check(args[2])) {
        const long layer = PyLong_AsLong(args[0]);
        const long heads = PyLong_AsLong(args[3]);
        const double scale = PyFloat_AsDouble(args[4]);
        const long factor = PyLong_AsLong(args[5]);
        const int rl = nargs == 7 ? PyObject_IsTrue(args[6]) : 1;
        if (PyErr_Occurred()) {
            PyErr_Clear();                                   // odd argument types: let Python complain
        } else if (layer >= 0 && layer < static_cast<long>(self->cache->size())) {
            const int st = record_steady(self, layer, args[0], args[1], args[2], heads, scale, factor, rl);
            if (st < 0) return nullptr;
            if (st > 0) Py_RETURN_NONE;
        }
    }
    if (!self->slow_cb) { PyErr_SetString(PyExc_RuntimeError, "recorder is closed"); return nullptr; }
    return PyObject_Vectorcall(self->slow_cb, args, nargs, kwnames);
}

// attend(layer, query, key, value, heads, scale, factor, round_logits, tapped) -> Tensor | None
// The processor's attention on daam_attend for a deferred trace: in the steady state (the call looks like the one the layer's
// DaamAttendDesc was built for) allocate the output, launch on torch's current stream, record Q / K for the batched tap --
// one C call instead of HeatMapEngine.attend's Python + ctypes (8 us -> the launch cost).  Everything else goes to the engine.
PyObject* Recorder_attend(Recorder* self, PyObject* const* args, Py_ssize_t nargs, PyObject* kwnames) {
    const bool has_kw = kwnames != nullptr && PyTuple_GET_SIZE(kwnames) > 0;
    if (!has_kw && nargs == 9 && self->attend_fn && self->native_ctx && THPVariable_Check(args[1]) && THPVariable_Check(args[2]) &&
        THPVariable_Check(args[3])) {
        const long layer = PyLong_AsLong(args[0]);
        const long heads = PyLong_AsLong(args[4]);
        const double scale = PyFloat_AsDouble(args[5]);
        const long factor = PyLong_AsLong(args[6]);
        const int rl = PyObject_IsTrue(args[7]);
        const int tapped = PyObject_IsTrue(args[8]);
        if (PyErr_Occurred()) {
            PyErr_Clear();
        } else if (layer >= 0 && layer < static_cast<long>(self->acache->size())) {
            bool steady = false;
            at::Tensor out;
            int rc = 0;
            try {
                const AttendCache& a = (*self->acache)[layer];
                const at::Tensor& q = THPVariable_Unpack(args[1]);
                const at::Tensor& k = THPVariable_Unpack(args[2]);
                const at::Tensor& v = THPVariable_Unpack(args[3]);
                const auto dev = q.device();
                steady = a.valid && same3(q, a.q_size) && same3(k, a.k_size) && same3(v, a.k_size) &&
                         static_cast<int>(q.scalar_type()) == a.dtype && static_cast<int>(k.scalar_type()) == a.dtype &&
                         static_cast<int>(v.scalar_type()) == a.dtype &&
                         static_cast<int>(dev.type()) == a.device_type && static_cast<int>(dev.index()) == a.device &&
                         k.device() == dev && v.device() == dev && heads == a.heads && scale == a.scale && rl == a.round_logits &&
                         q.is_contiguous() && k.is_contiguous() && v.is_contiguous() && q.has_storage() &&
                         !((q.requires_grad() || k.requires_grad() || v.requires_grad()) && c10::GradMode::is_enabled());
                if (steady) {
                    out = at::empty_like(q);
                    void* stream = c10::hip::getCurrentHIPStream(dev.index()).stream();
                    rc = self->attend_fn(self->native_ctx, static_cast<int>(layer), q.data_ptr(), k.data_ptr(), v.data_ptr(),
                                         out.data_ptr(), reinterpret_cast<const void*>(a.desc), 0, stream);
                }
            } catch (...) {
                steady = false;
            }
            if (steady && rc == 0) {
                if (tapped) {
                    const int st = record_steady(self, layer, args[0], args[1], args[2], heads, scale, factor, rl);
                    if (st < 0) return nullptr;
                    if (st == 0) {                           // the tap of this layer is not in ITS steady state yet
                        if (!self->slow_cb) { PyErr_SetString(PyExc_RuntimeError, "recorder is closed"); return nullptr; }
                        PyObject* targs[7] = {args[0], args[1], args[2], args[4], args[5], args[6], args[7]};
                        PyObject* r = PyObject_Vectorcall(self->slow_cb, targs, 7, nullptr);
                        if (!r) return nullptr;
                        Py_DECREF(r);
                    }
                }
                return THPVariable_Wrap(out);
            }
            // a library error (or DAAM_E_UNSUPPORTED for an unaligned view): the engine repeats the call and reports / declines
        }
    }
    if (!self->attend_slow_cb) { PyErr_SetString(PyExc_RuntimeError, "recorder has no attend path"); return nullptr; }
    return PyObject_Vectorcall(self->attend_slow_cb, args, nargs, kwnames);
}

// set_native(ctx_addr, daam_attend_addr, attend_slow): the engine's native context and entry point (0, 0: none)
PyObject* Recorder_set_native(Recorder* self, PyObject* args) {
    unsigned long long ctx, fn;
    PyObject* cb;
    if (!PyArg_ParseTuple(args, "KKO", &ctx, &fn, &cb)) return nullptr;
    self->native_ctx = reinterpret_cast<void*>(static_cast<uintptr_t>(ctx));
    self->attend_fn = reinterpret_cast<AttendFn>(static_cast<uintptr_t>(fn));
    Py_INCREF(cb);
    Py_XSETREF(self->attend_slow_cb, cb);
    Py_RETURN_NONE;
}

// set_attend_cache(layer, query, key, heads, scale, round_logits, desc_addr)
PyObject* Recorder_set_attend_cache(Recorder* self, PyObject* args) {
    int layer, rl;
    long heads;
    double scale;
    unsigned long long desc;
    PyObject *qo, *ko;
    if (!PyArg_ParseTuple(args, "iOOldpK", &layer, &qo, &ko, &heads, &scale, &rl, &desc)) return nullptr;
    if (layer < 0 || layer >= static_cast<int>(self->acache->size()) || !THPVariable_Check(qo) || !THPVariable_Check(ko)) {
        PyErr_SetString(PyExc_ValueError, "set_attend_cache: bad layer or tensors");
        return nullptr;
    }
    const at::Tensor& q = THPVariable_Unpack(qo);
    const at::Tensor& k = THPVariable_Unpack(ko);
    if (q.dim() != 3 || k.dim() != 3) { PyErr_SetString(PyExc_ValueError, "set_attend_cache: query / key must be 3-d"); return nullptr; }
    AttendCache& a = (*self->acache)[layer];
    for (int i = 0; i < 3; ++i) { a.q_size[i] = q.sizes()[i]; a.k_size[i] = k.sizes()[i]; }
    a.dtype = static_cast<int>(q.scalar_type());
    a.device_type = static_cast<int>(q.device().type());
    a.device = static_cast<int>(q.device().index());
    a.heads = heads; a.scale = scale; a.round_logits = rl; a.desc = desc;
    a.valid = desc != 0;
    Py_RETURN_NONE;
}

// set_cache(layer, query, key, heads, scale, factor, round_logits, desc_addr)
PyObject* Recorder_set_cache(Recorder* self, PyObject* args) {
    int layer, rl;
    long heads, factor;
    double scale;
    unsigned long long desc;
    PyObject *qo, *ko;
    if (!PyArg_ParseTuple(args, "iOOldlpK", &layer, &qo, &ko, &heads, &scale, &factor, &rl, &desc)) return nullptr;
    if (layer < 0 || layer >= static_cast<int>(self->cache->size()) || !THPVariable_Check(qo) || !THPVariable_Check(ko)) {
        PyErr_SetString(PyExc_ValueError, "set_cache: bad layer or tensors");
        return nullptr;
    }
    const at::Tensor& q = THPVariable_Unpack(qo);
    const at::Tensor& k = THPVariable_Unpack(ko);
    if (q.dim() != 3 || k.dim() != 3) { PyErr_SetString(PyExc_ValueError, "set_cache: query / key must be 3-d"); return nullptr; }
    LayerCache& c = (*self->cache)[layer];
    for (int i = 0; i < 3; ++i) { c.q_size[i] = q.sizes()[i]; c.k_size[i] = k.sizes()[i]; }
    c.dtype = static_cast<int>(q.scalar_type());
    c.device_type = static_cast<int>(q.device().type());
    c.device = static_cast<int>(q.device().index());
    c.heads = heads; c.factor = factor; c.scale = scale; c.round_logits = rl; c.desc = desc;
    c.valid = true;
    Py_RETURN_NONE;
}

// record(layer, query, key, desc_addr): unconditional (the engine validated the call)
PyObject* Recorder_record(Recorder* self, PyObject* args) {
    int layer;
    unsigned long long desc;
    PyObject *qo, *ko;
    if (!PyArg_ParseTuple(args, "iOOK", &layer, &qo, &ko, &desc)) return nullptr;
    if (layer < 0 || layer >= static_cast<int>(self->cache->size()) || !THPVariable_Check(qo) || !THPVariable_Check(ko)) {
        PyErr_SetString(PyExc_ValueError, "record: bad layer or tensors");
        return nullptr;
    }
    try {
        push(self, layer, THPVariable_Unpack(qo), THPVariable_Unpack(ko), desc);
    } catch (const std::exception& e) {
        PyErr_SetString(PyExc_RuntimeError, e.what());
        return nullptr;
    }
    Py_RETURN_NONE;
}

PyObject* Recorder_pending(Recorder* self, PyObject* arg) {            // recorded steps of one layer
    const long layer = PyLong_AsLong(arg);
    if (layer == -1 && PyErr_Occurred()) return nullptr;
    if (layer < 0 || layer >= static_cast<long>(self->cnt->size())) return PyLong_FromLong(0);
    return PyLong_FromLong((*self->cnt)[layer]);
}

PyObject* Recorder_count(Recorder* self, PyObject*) { return PyLong_FromSize_t(self->layers->size()); }

// buffers() -> (n, layers_addr, q_addr, k_addr, desc_addr): valid until the next record / drop
PyObject* Recorder_buffers(Recorder* self, PyObject*) {
    return Py_BuildValue("(nKKKK)", static_cast<Py_ssize_t>(self->layers->size()),
                         static_cast<unsigned long long>(reinterpret_cast<uintptr_t>(self->layers->data())),
                         static_cast<unsigned long long>(reinterpret_cast<uintptr_t>(self->qp->data())),
                         static_cast<unsigned long long>(reinterpret_cast<uintptr_t>(self->kp->data())),
                         static_cast<unsigned long long>(reinterpret_cast<uintptr_t>(self->dp->data())));
}

PyObject* Recorder_drop(Recorder* self, PyObject*) {                   // forget the recorded taps
    self->layers->clear(); self->qp->clear(); self->kp->clear(); self->dp->clear();
    if (self->keep->size() >= 64 && reaper().usable()) {
        std::vector<c10::Storage> batch;
        batch.reserve(self->keep->capacity());
        batch.swap(*self->keep);
        Py_BEGIN_ALLOW_THREADS                                  // may wait for the previous batch
        reaper().submit(std::move(batch));
        Py_END_ALLOW_THREADS
    } else {
        self->keep->clear();
    }
    self->held = 0;
    std::fill(self->cnt->begin(), self->cnt->end(), 0);
    Py_RETURN_NONE;
}

PyObject* Recorder_reset_touched(Recorder* self, PyObject*) {
    std::fill(self->touched->begin(), self->touched->end(), 0);
    Py_RETURN_NONE;
}

PyObject* Recorder_invalidate(Recorder* self, PyObject*) {             // forget every layer's cached call shape
    for (auto& c : *self->cache) c.valid = false;
    for (auto& a : *self->acache) a.valid = false;
    Py_RETURN_NONE;
}

PyObject* Recorder_set_window(Recorder* self, PyObject* arg) {
    const long w = PyLong_AsLong(arg);
    if (w == -1 && PyErr_Occurred()) return nullptr;
    self->window = static_cast<int>(w < 1 ? 1 : w);
    Py_RETURN_NONE;
}

PyObject* Recorder_get_window(Recorder* self, PyObject*) { return PyLong_FromLong(self->window); }

PyObject* Recorder_set_budget(Recorder* self, PyObject* arg) {
    const long long b = PyLong_AsLongLong(arg);
    if (b == -1 && PyErr_Occurred()) return nullptr;
    self->budget = b <= 0 ? INT64_MAX : b;
    Py_RETURN_NONE;
}

PyObject* Recorder_full(Recorder* self, PyObject* arg) {               // would tap(layer) launch first?
    const long layer = PyLong_AsLong(arg);
    if (layer == -1 && PyErr_Occurred()) return nullptr;
    return PyBool_FromLong(layer >= 0 && layer < static_cast<long>(self->cnt->size()) && must_launch(self, layer));
}

PyObject* Recorder_held_bytes(Recorder* self, PyObject*) { return PyLong_FromLongLong(self->held); }

PyMethodDef Recorder_methods[] = {
    {"tap", reinterpret_cast<PyCFunction>(reinterpret_cast<void (*)()>(Recorder_tap)), METH_FASTCALL | METH_KEYWORDS,
     "tap(layer, query, key, heads, scale, factor, round_logits=True): record one deferred tap"},
    {"attend", reinterpret_cast<PyCFunction>(reinterpret_cast<void (*)()>(Recorder_attend)), METH_FASTCALL | METH_KEYWORDS,
     "attend(layer, query, key, value, heads, scale, factor, round_logits, tapped): daam_attend + record the deferred tap"},
    {"set_native", reinterpret_cast<PyCFunction>(Recorder_set_native), METH_VARARGS, "native context, daam_attend address, Python attend path"},
    {"set_attend_cache", reinterpret_cast<PyCFunction>(Recorder_set_attend_cache), METH_VARARGS, "remember a layer's validated attend call"},
    {"set_cache", reinterpret_cast<PyCFunction>(Recorder_set_cache), METH_VARARGS, "remember a layer's validated call shape"},
    {"record", reinterpret_cast<PyCFunction>(Recorder_record), METH_VARARGS, "record a tap the engine validated"},
    {"pending", reinterpret_cast<PyCFunction>(Recorder_pending), METH_O, "recorded steps of one layer"},
    {"count", reinterpret_cast<PyCFunction>(Recorder_count), METH_NOARGS, "recorded taps"},
    {"buffers", reinterpret_cast<PyCFunction>(Recorder_buffers), METH_NOARGS, "(n, layers, q, k, desc) host addresses"},
    {"drop", reinterpret_cast<PyCFunction>(Recorder_drop), METH_NOARGS, "forget the recorded taps"},
    {"reset_touched", reinterpret_cast<PyCFunction>(Recorder_reset_touched), METH_NOARGS, ""},
    {"invalidate", reinterpret_cast<PyCFunction>(Recorder_invalidate), METH_NOARGS, ""},
    {"set_window", reinterpret_cast<PyCFunction>(Recorder_set_window), METH_O, "steps per layer before tap() asks for a flush"},
    {"get_window", reinterpret_cast<PyCFunction>(Recorder_get_window), METH_NOARGS, ""},
    {"set_budget", reinterpret_cast<PyCFunction>(Recorder_set_budget), METH_O, "bytes of recorded Q / K before tap() asks for a flush"},
    {"full", reinterpret_cast<PyCFunction>(Recorder_full), METH_O, "would tap(layer) launch before recording?"},
    {"held_bytes", reinterpret_cast<PyCFunction>(Recorder_held_bytes), METH_NOARGS, ""},
    {nullptr, nullptr, 0, nullptr}};

PyTypeObject RecorderType = {PyVarObject_HEAD_INIT(nullptr, 0)};

PyObject* module_drain(PyObject*, PyObject*) {                         // every released storage is back in the allocator
    Py_BEGIN_ALLOW_THREADS
    reaper().drain();
    Py_END_ALLOW_THREADS
    Py_RETURN_NONE;
}

PyObject* module_shutdown(PyObject*, PyObject*) {                      // registered with atexit by daam_amd.engine
    Py_BEGIN_ALLOW_THREADS
    reaper().shutdown();
    Py_END_ALLOW_THREADS
    Py_RETURN_NONE;
}

PyObject* module_set_sync_release(PyObject*, PyObject* arg) {          // True: drop() frees inline (DAAM_SYNC_RELEASE=1)
    const int on = PyObject_IsTrue(arg);
    if (on < 0) return nullptr;
    if (on) {
        Py_BEGIN_ALLOW_THREADS
        reaper().drain();
        Py_END_ALLOW_THREADS
    }
    reaper().set_sync(on != 0);
    Py_RETURN_NONE;
}

PyMethodDef module_methods[] = {
    {"drain", module_drain, METH_NOARGS, "wait until the storages released by Recorder.drop() are back in the allocator"},
    {"shutdown", module_shutdown, METH_NOARGS, "drain and stop the release thread; later drops free inline"},
    {"set_sync_release", module_set_sync_release, METH_O, "free recorded storages inline instead of on the release thread"},
    {nullptr, nullptr, 0, nullptr}};

PyModuleDef module_def = {PyModuleDef_HEAD_INIT, "_fastpath", "host-side recorder of deferred DAAM taps", -1, module_methods};

}  // namespace

PyMODINIT_FUNC PyInit__fastpath(void) {
    RecorderType.tp_name = "daam_amd._fastpath.Recorder";
    RecorderType.tp_basicsize = sizeof(Recorder);
    RecorderType.tp_flags = Py_TPFLAGS_DEFAULT | Py_TPFLAGS_HAVE_GC;     // engine <-> recorder callbacks form a cycle
    RecorderType.tp_traverse = reinterpret_cast<traverseproc>(Recorder_traverse);
    RecorderType.tp_clear = reinterpret_cast<inquiry>(Recorder_clear);
    RecorderType.tp_new = PyType_GenericNew;
    RecorderType.tp_init = reinterpret_cast<initproc>(Recorder_init);
    RecorderType.tp_dealloc = reinterpret_cast<destructor>(Recorder_dealloc);
    RecorderType.tp_methods = Recorder_methods;
    if (PyType_Ready(&RecorderType) < 0) return nullptr;
    pthread_atfork(nullptr, nullptr, [] { reaper().disable_after_fork(); });
    PyObject* m = PyModule_Create(&module_def);
    if (!m) return nullptr;
    Py_INCREF(&RecorderType);
    if (PyModule_AddObject(m, "Recorder", reinterpret_cast<PyObject*>(&RecorderType)) < 0) {
        Py_DECREF(&RecorderType);
        Py_DECREF(m);
        return nullptr;
    }
    return m;
}
