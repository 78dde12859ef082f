// tac_ext.cpp — compiled PyTorch binding of the hot call (round 5): the fused Melspectrogram chain as ONE foreign call from the
// layer that ends it.
//
// The boundary stays the C ABI of include/tac_amd.h (libtac_amd.so; raw pointers, sizes, a hipStream_t); ctypes (_native.py /
// _hip.py) stays the general binding — every op, every route, the tests.  What this module adds is the steady state of the
// reference idiom nn.Sequential(*Melspectrogram(...), AmplitudeToDb()) (reference layers.py:84-102, 307-347) without per-call
// Python conversions: a MelPlan holds everything one (waveform layout, window, filterbank, parameters) combination needs — the
// entry point's address, the packed bank, the geometry descriptor, the output shape — and launch(wave) checks the layout and the
// stamps of the constants, allocates the output, reads torch's current HIP stream and calls tac_melspec_sparse_f32.  The same
// launch is registered as the dispatcher op tac_amd::melspec_planned (CUDA + Meta kernels), so that profilers and dispatch modes
// see an op; the layers call the plan directly when no mode is active.
//
// Host code only (the kernels live in libtac_amd.so): built with g++ against the torch headers by csrc/Makefile.
#include <torch/extension.h>
#include <frameobject.h>
#include <ATen/record_function.h>
#include <c10/hip/HIPFunctions.h>
#include <c10/hip/HIPGuard.h>
#include <c10/hip/HIPStream.h>

#include <atomic>
#include <cstdint>
#include <memory>
#include <mutex>
#include <vector>

namespace {

// int tac_melspec_sparse_f32(wave, window, desc, power, wpack, desc_dev, info_host, n_mels, db, db_ref, db_amin, out, stream)
typedef int (*melspec_fn)(const float*, const float*, const void*, float, const float*, const int32_t*, const int32_t*, int32_t,
                          int, float, float, float*, void*);

std::atomic<int64_t> g_epoch{0};            // mirrors _hip._epoch (invalidate() without arguments)

struct MelPlan {
    melspec_fn fn = nullptr;
    at::Tensor window, fb, wpack, dsc;       // kept alive with the plan; window / fb also watched (version, address)
    std::vector<char> desc;                  // tac_stft_desc, by value
    std::vector<int32_t> info;               // info_host[8]
    float power = 2.0f, ref = 1.0f, amin = 1e-7f;
    int32_t n_mels = 0;
    int db = 0;
    std::vector<int64_t> out_shape, wave_sizes, wave_strides;
    c10::DeviceIndex dev = 0;
    uint32_t win_version = 0, fb_version = 0;
    const void* win_ptr = nullptr;
    const void* fb_ptr = nullptr;
    int64_t epoch = 0;
    std::atomic<int64_t> launches{0};        // (plans are shared between threads)
    std::atomic<int> last_rc{0};             // TAC_* code of the last refused launch (0: none)

    // (the waveform's device need not be the CURRENT one: launch() switches to it for the call)
    bool matches(const at::Tensor& wave) const {
        return wave.scalar_type() == at::kFloat && wave.is_cuda() && wave.device().index() == dev && wave.sizes() == wave_sizes &&
               wave.strides() == wave_strides && epoch == g_epoch.load(std::memory_order_relaxed) &&
               window._version() == win_version && window.data_ptr() == win_ptr && fb._version() == fb_version &&
               fb.data_ptr() == fb_ptr;
    }

    // the (.., n_mels, frames) view of a fresh frame-major buffer, or an undefined tensor when the plan does not apply (any more)
    // or the launch was refused (last_rc says with which code)
    at::Tensor launch(const at::Tensor& wave) {
        if (!matches(wave)) return at::Tensor();
        RECORD_USER_SCOPE("tac_amd::melspectrogram (planned)");
        const c10::hip::HIPGuard device_guard(dev);
        at::Tensor out = at::empty(out_shape, wave.options());
        void* stream = c10::hip::getCurrentHIPStream(dev).stream();
        const int rc = fn(wave.data_ptr<float>(), static_cast<const float*>(win_ptr), desc.data(), power, wpack.data_ptr<float>(),
                          dsc.data_ptr<int32_t>(), info.data(), n_mels, db, ref, amin, out.data_ptr<float>(), stream);
        if (rc != 0) {
            last_rc.store(rc, std::memory_order_relaxed);
            return at::Tensor();
        }
        launches.fetch_add(1, std::memory_order_relaxed);
        return out.transpose(-2, -1);
    }
};

std::mutex g_mutex;
std::vector<std::shared_ptr<MelPlan>> g_plans;      // ids handed to the dispatcher op

int64_t register_plan(const std::shared_ptr<MelPlan>& p) {
    std::lock_guard<std::mutex> lock(g_mutex);
    for (size_t i = 0; i < g_plans.size(); ++i)      // an id drop_plan freed is handed out again: the table does not grow with time
        if (!g_plans[i]) {
            g_plans[i] = p;
            return (int64_t)i;
        }
    g_plans.push_back(p);
    return (int64_t)g_plans.size() - 1;
}

std::shared_ptr<MelPlan> plan_by_id(int64_t id) {
    std::lock_guard<std::mutex> lock(g_mutex);
    TORCH_CHECK(id >= 0 && id < (int64_t)g_plans.size() && g_plans[id], "tac_amd::melspec_planned: unknown plan ", id);
    return g_plans[id];
}

at::Tensor melspec_planned_cuda(const at::Tensor& wave, int64_t plan) {
    at::Tensor y = plan_by_id(plan)->launch(wave);
    TORCH_CHECK(y.defined(), "tac_amd::melspec_planned: the plan no longer matches this call (layout, device, window or "
                             "filterbank changed) or the launch was refused");
    return y;
}

at::Tensor melspec_planned_meta(const at::Tensor& wave, int64_t plan) {
    return at::empty(plan_by_id(plan)->out_shape, wave.options()).transpose(-2, -1);
}

}  // namespace

TORCH_LIBRARY_FRAGMENT(tac_amd, m) {
    m.def("melspec_planned(Tensor wave, int plan) -> Tensor");
}
TORCH_LIBRARY_IMPL(tac_amd, CUDA, m) {
    m.impl("melspec_planned", &melspec_planned_cuda);
}
TORCH_LIBRARY_IMPL(tac_amd, Meta, m) {
    m.impl("melspec_planned", &melspec_planned_meta);
}

// ---- _lazy.ends_chain in C (round 6): is `module`, asked from inside its forward(), the LAST child of the nn.Sequential that called
// it — through any nesting of such containers — with the outermost of them called by something that is not a container?  The Python
// form of the same walk (sys._getframe, f_locals) costs ~1.5 us per layer; this one reads the frames' code objects and first
// local directly (CPython 3.8 - 3.10 frame layout; other interpreters report -1 and the Python walk answers).
PyObject* g_seq_forward_code = nullptr;     // nn.Sequential.forward.__code__
PyObject* g_module_py = nullptr;            // co_filename of nn.Module.__call__ (the _call_impl / _wrapped_call_impl frames)
PyObject* g_str_modules = nullptr;          // interned "_modules", "_tac_realizes"
PyObject* g_str_realizes = nullptr;

int chain_end(PyObject* module, int skip) {
#if PY_VERSION_HEX >= 0x030B0000 || PY_VERSION_HEX < 0x03080000
    (void)module;
    (void)skip;
    return -1;
#else
    if (!g_seq_forward_code || !g_module_py) return -1;
    PyFrameObject* f = PyEval_GetFrame();               // the Python frame this function was called from (borrowed): the layer's
    for (int i = 0; f && i < skip; ++i) f = f->f_back;  // forward(), or `skip` wrappers below it
    if (!f) return -1;
    f = f->f_back;                                      // forward() was called by nn.Module._call_impl
    PyObject* child = module;
    for (;;) {
        while (f && (f->f_code->co_filename == g_module_py || PyUnicode_Compare(f->f_code->co_filename, g_module_py) == 0)) f = f->f_back;
        if (!f || (PyObject*)f->f_code != g_seq_forward_code) return child != module ? 1 : 0;
        PyObject* cont = f->f_code->co_argcount > 0 ? f->f_localsplus[0] : nullptr;     // `self` of Sequential.forward
        if (!cont) return 0;
        PyObject** dp = _PyObject_GetDictPtr(cont);
        PyObject* mods = (dp && *dp) ? PyDict_GetItem(*dp, g_str_modules) : nullptr;      // borrowed
        if (!mods || !PyDict_Check(mods) || PyDict_Size(mods) == 0) return 0;
        if (_PyType_Lookup(Py_TYPE(cont), g_str_realizes)) return 0;                      // a factory container realises by itself
        PyObject *key, *value, *last = nullptr;
        Py_ssize_t pos = 0;
        while (PyDict_Next(mods, &pos, &key, &value)) last = value;                      // (insertion order; a handful of children)
        if (last != child) return 0;
        child = cont;
        f = f->f_back;
    }
#endif
}

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m) {
    m.def("chain_end_init", [](py::object seq_forward_code, py::object module_py) {
        Py_XDECREF(g_seq_forward_code);
        Py_XDECREF(g_module_py);
        g_seq_forward_code = seq_forward_code.inc_ref().ptr();
        g_module_py = module_py.inc_ref().ptr();
        if (!g_str_modules) g_str_modules = PyUnicode_InternFromString("_modules");
        if (!g_str_realizes) g_str_realizes = PyUnicode_InternFromString("_tac_realizes");
    });
    m.def("chain_end", [](py::handle module, int skip) { return chain_end(module.ptr(), skip); });
    // called straight from a layer's forward(): True / False (only offered where chain_end answers: chain_end_supported)
    m.def("ends_chain", [](py::handle module) { return chain_end(module.ptr(), 0) == 1; });
    m.attr("chain_end_supported") = (PY_VERSION_HEX < 0x030B0000 && PY_VERSION_HEX >= 0x03080000);
    m.doc() = "compiled binding of the fused Melspectrogram launch (C ABI of libtac_amd.so behind it)";
    m.attr("ABI") = 1;
    m.def("set_epoch", [](int64_t e) { g_epoch.store(e, std::memory_order_relaxed); });
    m.def("drop_plan", [](int64_t id) {
        std::lock_guard<std::mutex> lock(g_mutex);
        if (id >= 0 && id < (int64_t)g_plans.size()) g_plans[id].reset();
    });
    py::class_<MelPlan, std::shared_ptr<MelPlan>>(m, "MelPlan")
        .def(py::init([](uint64_t fn_address, at::Tensor wave, at::Tensor window, at::Tensor fb, at::Tensor wpack, at::Tensor dsc,
                         py::bytes desc, std::vector<int32_t> info, double power, int64_t n_mels, bool db, double ref, double amin,
                         std::vector<int64_t> out_shape, int64_t epoch) {
            TORCH_CHECK(fn_address != 0, "null entry point");
            TORCH_CHECK(wave.is_cuda() && wave.scalar_type() == at::kFloat && window.scalar_type() == at::kFloat &&
                            fb.scalar_type() == at::kFloat && wpack.scalar_type() == at::kFloat && dsc.scalar_type() == at::kInt,
                        "MelPlan: float32 device tensors expected");
            auto p = std::make_shared<MelPlan>();
            p->fn = reinterpret_cast<melspec_fn>(fn_address);
            p->window = window;
            p->fb = fb;
            p->wpack = wpack;
            p->dsc = dsc;
            const std::string d = desc;
            p->desc.assign(d.begin(), d.end());
            TORCH_CHECK(info.size() == 8, "info_host has eight entries");
            p->info = std::move(info);
            p->power = (float)power;
            p->n_mels = (int32_t)n_mels;
            p->db = db ? 1 : 0;
            p->ref = (float)ref;
            p->amin = (float)amin;
            p->out_shape = std::move(out_shape);
            p->wave_sizes = wave.sizes().vec();
            p->wave_strides = wave.strides().vec();
            p->dev = wave.device().index();
            p->win_version = window._version();
            p->fb_version = fb._version();
            p->win_ptr = window.data_ptr();
            p->fb_ptr = fb.data_ptr();
            p->epoch = epoch;
            return p;
        }))
        .def("launch", [](MelPlan& p, const at::Tensor& wave) -> py::object {
            at::Tensor y = p.launch(wave);
            if (!y.defined()) return py::none();
            return py::cast(y);
        })
        .def("matches", &MelPlan::matches)
        .def("register", [](std::shared_ptr<MelPlan> p) { return register_plan(p); })
        .def_property_readonly("launches", [](const MelPlan& p) { return p.launches.load(); })
        .def_property_readonly("last_rc", [](const MelPlan& p) { return p.last_rc.load(); })
        .def_property_readonly("window", [](const MelPlan& p) { return p.window; })
        .def_property_readonly("fb", [](const MelPlan& p) { return p.fb; });
}
