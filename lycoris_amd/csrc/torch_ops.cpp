// torch_ops.cpp -- PyTorch-ROCm custom ops over the C ABI (include/lycoris_amd.h): TORCH_LIBRARY(lycoris_amd).
//
// SURVEY 8b "Native C-ABI layer to export": one forward + one backward schema per kernel family, autograd glue in C++
// (torch::autograd::Function registered on the Autograd key), Meta kernels for FakeTensor / torch.compile tracing.
// The reference's per-layer host work is a chain of ATen ops driven from Python (lycoris/modules/lokr.py:543-566);
// here a layer's forward is ONE dispatcher call and its backward ONE autograd node, all host work in C++:
// no ctypes, no Python autograd.Function, no per-call Python allocation (~185 us -> tens of us per layer fwd+bwd).
//
// Host-only translation unit (g++): the kernels live in liblycoris_amd.so, this file only passes device pointers,
// sizes and torch's current HIP stream through `extern "C"` calls.
#include <ATen/ATen.h>
#include <ATen/autocast_mode.h>
#include <c10/core/DeviceGuard.h>
#include <c10/hip/HIPGuard.h>
#include <c10/hip/HIPStream.h>
#include <torch/csrc/autograd/custom_function.h>
#include <torch/extension.h>
#include <ATen/SavedTensorHooks.h>
#include <torch/library.h>

#include <hip/hip_runtime_api.h>
#include <torch/csrc/autograd/engine.h>

#include <algorithm>
#include <atomic>
#include <mutex>
#include <unordered_map>
#include <vector>

#include "../../include/lycoris_amd.h"

namespace {

using at::Tensor;
using torch::autograd::AutogradContext;
using torch::autograd::variable_list;

constexpr int F32_ROWS = 0x100;

int dtype_code(at::ScalarType t) {
  switch (t) {
    case at::kFloat: return LYC_F32;
    case at::kHalf: return LYC_F16;
    case at::kBFloat16: return LYC_BF16;
    default: TORCH_CHECK(false, "lycoris_amd supports float32/float16/bfloat16 activations, got ", t);
  }
}

// A/B and regression-test switch (torch.ops.lycoris_amd.locon_reg_staged): the rank-r launches stay on the register-staged kernel of
// rounds 1-5 (LYC_BNECK_REG in the dtype argument of the C ABI) instead of the LDS-DMA kernel of round 6
bool g_locon_reg = false;
inline int lc(int code) { return g_locon_reg ? (code | LYC_BNECK_REG) : code; }

void* stream_of(const Tensor& t) { return c10::hip::getCurrentHIPStream(t.device().index()).stream(); }

void check_rc(int rc, const char* what) { TORCH_CHECK(rc == 0, what, " failed (code ", rc, "): ", lyc_last_error()); }

void require_device(const Tensor& t, const char* what) {
  TORCH_CHECK(t.is_cuda(), "lycoris_amd: ", what, " is on ", t.device(),
              "; the adapter hot path only runs on the MI355X HIP device (there is no CPU fallback by design).");
}

// fp32 contiguous, detached view of a (small) factor
Tensor f32c(const Tensor& t) {
  if (t.scalar_type() == at::kFloat && t.is_contiguous()) return t;  // the usual case: no new TensorImpl (host time per layer)
  Tensor f = t.detach();
  if (f.scalar_type() != at::kFloat) f = f.to(at::kFloat);
  return f.contiguous();
}

// dx [M, I] in the shape of the activation it belongs to; no new TensorImpl when that already is 2-D
Tensor shaped_like(const Tensor& dx, const Tensor& x) { return x.dim() == 2 ? dx : dx.view(x.sizes()); }

const void* cptr(const Tensor& t) { return t.defined() ? t.const_data_ptr() : nullptr; }
void* mptr(const Tensor& t) { return t.defined() ? t.mutable_data_ptr() : nullptr; }
const float* cfp(const Tensor& t) { return t.defined() ? t.const_data_ptr<float>() : nullptr; }
float* mfp(const Tensor& t) { return t.defined() ? t.mutable_data_ptr<float>() : nullptr; }

Tensor rows_of(const Tensor& x, int64_t feat) {
  if (x.dim() == 2 && x.size(1) == feat && x.is_contiguous()) return x;
  Tensor r = x.reshape({-1, feat});
  return r.is_contiguous() ? r : r.contiguous();
}

// ---- fused gradient accumulation (mirror of ops.fused_grad_accumulation) ------------------------------------------------
// When a factor is a leaf whose .grad exists as a contiguous fp32 tensor (e.g. a view of grad_sync's arena) the backward
// kernels accumulate straight into it and autograd gets an undefined gradient for that input.  `callback` (a Python
// callable, e.g. AdapterGradSync._on_grad_ready) is told which parameter was updated.
struct Accum {
  bool enabled = false;
  py::object* callback = nullptr;  // leaked on purpose: must not be destroyed after the interpreter has shut down
  bool has_callback = false;       // callback set and not None (readable without the GIL)
  py::object* batch_callback = nullptr;  // optional: called once with a LIST of parameters (notify_many)
  bool has_batch = false;
  std::mutex mu;
  // A parameter used by several layer calls of one forward pass (a shared module) gets one accumulation per call, but the
  // callback's contract is the autograd hook's: ONE report per parameter, after its LAST accumulation (the DP sync
  // counted a shared parameter twice and all-reduced its bucket early).  `expect()` counts the backward nodes created for a
  // parameter (forward, grad mode on); `notify()` reports when the count returns to zero.  A forward whose backward never runs
  // leaves a stale count: the report is then skipped, AdapterGradSync.finish() launches such buckets, and
  // reset_use_counts() (called by AdapterGradSync.zero_grad() / finish()) clears the map once per step.
  std::unordered_map<const void*, int> uses;
  // autograd runs a leaf's AccumulateGrad node -- and with it the tensor's post-accumulate-grad hooks -- even when the backward
  // node handed it an UNDEFINED gradient (torch >= 2.x: the hook of AdapterGradSync fires for every parameter of a
  // loss.backward()).  A parameter the kernels report through notify() must not be counted a second time by that hook:
  // `done_task` remembers the graph task (backward pass) in which a parameter was reported, fused_reports() tells the hook.
  std::unordered_map<const void*, int> done_task;
} g_accum;

// torch::autograd::Function::apply runs forward() with grad mode OFF: whether a backward node is being built is only visible to
// the wrapper that calls apply().  Every such wrapper records it here first (thread-local: apply() runs on the calling thread).
thread_local bool tl_grad_at_apply = false;
struct GradAtApply {
  GradAtApply() { tl_grad_at_apply = c10::GradMode::is_enabled(); }
};

// true when the backward node of a layer call on activation `x` will accumulate into factor.grad and notify() for it
// (the condition of grad_target() / finish_grad() below)
bool will_report(const Tensor& factor, const Tensor& x, bool cl = false) {
  if (!g_accum.enabled || !g_accum.has_callback || !factor.defined() || !factor.is_leaf() || !factor.requires_grad()) return false;
  if (!tl_grad_at_apply || !x.is_cuda()) return false;
  const c10::DispatchKeySet ks = x.key_set();
  if (ks.has(c10::DispatchKey::Python) || ks.has(c10::DispatchKey::Meta) || ks.has(c10::DispatchKey::Functionalize)) return false;
  const Tensor& gr = factor.grad();
  if (!gr.defined() || gr.scalar_type() != at::kFloat || gr.device() != factor.device()) return false;
  return cl ? gr.permute({0, 2, 3, 1}).is_contiguous() : gr.is_contiguous();  // cl: cl_grad_target()'s condition
}
// A forward that runs INSIDE a backward pass under saved-tensor hooks is the recomputation of non-reentrant activation checkpointing
// (torch.utils.checkpoint(use_reentrant=False), the diffusers default): its nodes only re-create the saved tensors of the ORIGINAL
// nodes and never run backward themselves, so they must not be counted (the counts never reached 0, no report fired from
// inside the backward pass and every bucket went out in finish() -- the DP overlap was silently lost).  Reentrant checkpointing
// recomputes inside a backward pass too, but without saved-tensor hooks, and runs the recomputed nodes in a nested pass: counted.
// (Both at once -- reentrant checkpointing inside a user's own saved_tensors_hooks -- is taken for the first case: a parameter shared
// by two layer calls is then reported once per call, which AdapterGradSync refuses loudly instead of reducing early in silence.)
bool inside_checkpoint_recompute() {
  return torch::autograd::get_current_graph_task_id() >= 0 && at::SavedTensorDefaultHooks::is_enabled() &&
         at::SavedTensorDefaultHooks::get_hooks().has_value();
}
void expect(const Tensor& param, const Tensor& x, bool cl = false) {
  if (!will_report(param, x, cl) || inside_checkpoint_recompute()) return;
  std::lock_guard<std::mutex> lk(g_accum.mu);
  ++g_accum.uses[param.unsafeGetTensorImpl()];
}

void notify(const Tensor& param) {
  if (g_accum.callback == nullptr || !param.defined()) return;
  {
    std::lock_guard<std::mutex> lk(g_accum.mu);
    auto it = g_accum.uses.find(param.unsafeGetTensorImpl());
    if (it != g_accum.uses.end()) {
      if (--it->second > 0) return;  // another layer call of this pass still has to accumulate into the same .grad
      g_accum.uses.erase(it);
    }
    g_accum.done_task[param.unsafeGetTensorImpl()] = torch::autograd::get_current_graph_task_id();
  }
  py::gil_scoped_acquire gil;
  if (!g_accum.callback->is_none()) (*g_accum.callback)(param);
}

// for the autograd hook of the same owner: true when the kernels report (or have reported, in the running backward pass) this
// parameter themselves -- its AccumulateGrad ran on an undefined gradient and must not be counted
bool fused_reports(const Tensor& param) {
  if (!param.defined() || !g_accum.enabled || !g_accum.has_callback) return false;
  std::lock_guard<std::mutex> lk(g_accum.mu);
  const void* key = param.unsafeGetTensorImpl();
  auto u = g_accum.uses.find(key);
  if (u != g_accum.uses.end() && u->second > 0) return true;  // parked / still to run: the report comes later in this pass
  auto d = g_accum.done_task.find(key);
  if (d == g_accum.done_task.end()) return false;
  const int task = torch::autograd::get_current_graph_task_id();
  return task >= 0 && d->second == task;
}

// the same for a batch of parameters (the end-of-backward flush of the parked layers): ONE call into Python with the list of those
// whose last accumulation this was, when the owner registered a batch callback (1576 calls of ~1.5 us each per SDXL backward otherwise)
void notify_many(const std::vector<Tensor>& params) {
  if (g_accum.callback == nullptr) return;
  if (g_accum.batch_callback == nullptr || !g_accum.has_batch) {
    for (const Tensor& p : params) notify(p);
    return;
  }
  std::vector<Tensor> done;
  done.reserve(params.size());
  {
    std::lock_guard<std::mutex> lk(g_accum.mu);
    for (const Tensor& p : params) {
      if (!p.defined()) continue;
      auto it = g_accum.uses.find(p.unsafeGetTensorImpl());
      if (it != g_accum.uses.end()) {
        if (--it->second > 0) continue;
        g_accum.uses.erase(it);
      }
      g_accum.done_task[p.unsafeGetTensorImpl()] = torch::autograd::get_current_graph_task_id();
      done.push_back(p);
    }
  }
  if (done.empty()) return;
  py::gil_scoped_acquire gil;
  if (!g_accum.batch_callback->is_none()) (*g_accum.batch_callback)(done);
}

// a real device tensor in eager mode (not a FakeTensor / functional wrapper seen while torch.compile traces)
bool eager_cuda(const Tensor& t) {
  const c10::DispatchKeySet ks = t.key_set();
  return t.is_cuda() && !ks.has(c10::DispatchKey::Python) && !ks.has(c10::DispatchKey::Meta) &&
         !ks.has(c10::DispatchKey::Functionalize);
}

// Fused accumulation follows the graph task (round 4; ADVICE r2 / VERDICT r3 #8): a factor's gradient is computed -- and added into
// `.grad` by the kernel -- only when THIS backward call needs it (ctx->needs_input_grad, which in C++ is per graph task).
// loss.backward() needs every leaf that requires grad; a gradient probe such as torch.autograd.grad(loss, [x]) needs none of the
// factors and leaves their `.grad` alone.  Callers that drive backward through autograd.grad and want the factor gradients list the
// factors among its inputs (allow_unused=True: the kernel has already added them into `.grad`, nothing is handed back).

// ---- saved variables that keep the identity of the parameters ---------------------------------------------------------------------
// Saved-tensor hooks hand the backward node COPIES of what the forward saved: non-reentrant activation checkpointing (the diffusers
// default) recomputes the forward and keeps `x.detach()` of every saved tensor, offloading hooks bring a fresh device copy.  For an
// activation that is all the backward needs.  For a factor it loses the Parameter: the copy has no `.grad` to accumulate into and
// is not the tensor the DP bucket counters know -- the gradient then went through AccumulateGrad, unreported, and its bucket was
// reduced in finish() instead of inside the backward pass (found by tests/test_gpu_grad_sync.py, round 4).  Leaf inputs that
// require grad (never the activation at index 0) are therefore also kept as plain references in saved_data, which the hooks do not
// see, and put back in place of their copies.
const std::string& leaf_key(size_t i) {  // no string building on the per-layer path
  static const std::string keys[8] = {"lyc_leaf0", "lyc_leaf1", "lyc_leaf2", "lyc_leaf3", "lyc_leaf4", "lyc_leaf5", "lyc_leaf6", "lyc_leaf7"};
  return keys[i < 8 ? i : 7];
}
void save_vars(AutogradContext* ctx, torch::autograd::variable_list vars) {
  TORCH_INTERNAL_ASSERT(vars.size() <= 8);
  for (size_t i = 1; i < vars.size(); ++i)
    if (vars[i].defined() && vars[i].is_leaf() && vars[i].requires_grad()) ctx->saved_data[leaf_key(i)] = vars[i];
  ctx->save_for_backward(std::move(vars));
}
torch::autograd::variable_list saved_vars(AutogradContext* ctx) {
  torch::autograd::variable_list s = ctx->get_saved_variables();
  for (size_t i = 1; i < s.size(); ++i) {
    auto it = ctx->saved_data.find(leaf_key(i));
    if (it == ctx->saved_data.end() || !it->second.isTensor()) continue;
    const Tensor& p = it->second.toTensor();
    if (p.defined() && s[i].defined() && !s[i].is_same(p) && s[i].sizes() == p.sizes()) s[i] = p;
  }
  return s;
}

// the tensor the kernel accumulates into: existing .grad (hand_back = false) or a fresh zero buffer (hand_back = true)
struct GradTarget {
  Tensor buf;
  bool hand_back = false;
};
GradTarget grad_target(const Tensor& factor, bool need, c10::optional<at::IntArrayRef> shape = c10::nullopt) {
  GradTarget g;
  if (!need) return g;
  if (g_accum.enabled && factor.is_leaf()) {
    const Tensor& gr = factor.grad();
    if (gr.defined() && gr.scalar_type() == at::kFloat && gr.is_contiguous() && gr.device() == factor.device()) {
      g.buf = gr;
      return g;
    }
  }
  g.buf = at::zeros(shape.has_value() ? *shape : factor.sizes(), factor.options().dtype(at::kFloat));
  g.hand_back = true;
  return g;
}
Tensor finish_grad(const Tensor& factor, const GradTarget& g) {
  if (!g.buf.defined()) return Tensor();
  if (g.hand_back) return g.buf.reshape(factor.sizes()).to(factor.scalar_type());
  notify(factor);
  return Tensor();
}

// torch.autocast parity (see ops._amp): an fp32 activation is cast to the autocast dtype, differentiably
Tensor amp(const Tensor& x) {
  if (x.is_cuda() && x.scalar_type() == at::kFloat && at::autocast::is_autocast_enabled(at::kCUDA))
    return x.to(at::autocast::get_autocast_dtype(at::kCUDA));
  return x;
}

// ---- deferred, grouped weight gradients (lyc_lokr_wgrad_group) --------------------------------------------------------------
// A factor gradient that is accumulated straight into `.grad` is invisible to autograd, so nothing in the backward pass waits
// for it.  The backward node of such a layer runs only its dx launch, parks (g, x, ws) here, and the engine's end-of-backward
// callback (or a full batch) hands all parked layers to ONE grouped launch per 24 layers: the per-layer launches were bound by
// the latency chain of a single resident round of workgroups, a batch is throughput-bound.  MI355X has the memory for it:
// the parked tensors of a whole SDXL backward are a few GB of 288.
struct DeferredLokr {
  Tensor g, x, f1, w1, w2, dw1, dw2, ws;
  int64_t M;
  int a, b, c, d, code;
  float alpha;
  void* stream;
  c10::DeviceIndex device;
  Tensor w2a, w2b, d_w2a, d_w2b;  // low-rank w2 = w2a @ w2b (all four defined): dw2 is then a slice of the flush's scratch arena
  Tensor w1a, w1b, d_w1a, d_w1b;  // decompose_both: w1 = w1a @ w1b; `f1` is the product, dw1 a slice of the scratch arena
};
struct DeferredLocon {
  Tensor g, x, t, dt, down, up, dd, du;  // dd / du: the .grad targets (either may be undefined)
  int64_t M;
  int I, O, r, code;
  float alpha;
  void* stream;
  c10::DeviceIndex device;
};
struct DeferredLokrConv {
  Tensor g_rows, x_rows, f1, w1, w2, dw1, dw2p, ws;  // dw2p: the window-major view of w2.grad
  int64_t B, H, W, dw1_blocks;
  int a, b, c, d, geom[8], code;
  float alpha;
  void* stream;
  c10::DeviceIndex device;
  Tensor w2a, w2b, d_w2a, d_w2b;  // low-rank w2 (all four defined): dw2p is then a slice of the flush's scratch arena
};
struct DeferredLoha {
  Tensor g, x, f[4], p[4], d[4];  // f: fp32 contiguous factors, p: the parameters (for the sync callback), d: .grad targets
  int64_t M;
  int I, O, r, code;
  float alpha;
  void* stream;
  c10::DeviceIndex device;
};
// One park list per DEVICE (a process that drives two GPUs has one autograd worker thread per device; they
// must not share a list, a lock or the "callback queued" state).  `enabled` / `flush_at` are process-wide settings.
struct DeferredLists {
  std::mutex mu;
  std::vector<DeferredLokr> lokr;
  std::vector<DeferredLocon> locon;
  std::vector<DeferredLoha> loha;
  std::vector<DeferredLokrConv> lokr_conv;
  bool callback_queued = false;
  int queued_task = -1;  // graph task the pending end-of-backward callback belongs to
  size_t size() const { return lokr.size() + locon.size() + loha.size() + lokr_conv.size(); }
};
constexpr int kMaxDevices = 64;
struct Deferred {
  std::atomic<bool> enabled{true};
  std::atomic<size_t> flush_at{48};
  DeferredLists dev[kMaxDevices];
} g_defer;
DeferredLists& lists_of(c10::DeviceIndex device) {
  TORCH_CHECK(device >= 0 && device < kMaxDevices, "lycoris_amd: device index ", (int)device, " out of range");
  return g_defer.dev[device];
}

// after launches on `stream`: whoever consumes .grad on the ambient stream must see them
void join_ambient(c10::DeviceIndex device, void* stream) {
  void* cur = c10::hip::getCurrentHIPStream(device).stream();
  if (cur == stream) return;
  hipEvent_t ev;
  TORCH_CHECK(hipEventCreateWithFlags(&ev, hipEventDisableTiming) == hipSuccess, "hipEventCreate failed");
  (void)hipEventRecord(ev, (hipStream_t)stream);
  (void)hipStreamWaitEvent((hipStream_t)cur, ev, 0);
  (void)hipEventDestroy(ev);
}

void flush_deferred(c10::DeviceIndex device) {
  std::vector<DeferredLokr> items;
  std::vector<DeferredLocon> litems;
  std::vector<DeferredLoha> hitems;
  std::vector<DeferredLokrConv> citems;
  {
    DeferredLists& L = lists_of(device);
    std::lock_guard<std::mutex> lock(L.mu);
    items.swap(L.lokr);
    litems.swap(L.locon);
    hitems.swap(L.loha);
    citems.swap(L.lokr_conv);
    L.callback_queued = false;
  }
  if (items.empty() && litems.empty() && hitems.empty() && citems.empty()) return;
  for (size_t lo = 0; lo < citems.size();) {  // Conv2d LoKr layers: one call per (stream, dtype) run
    size_t hi = lo + 1;
    while (hi < citems.size() && citems[hi].stream == citems[lo].stream && citems[hi].code == citems[lo].code) ++hi;
    const c10::DeviceGuard guard(c10::Device(c10::kCUDA, citems[lo].device));
    const c10::hip::HIPStreamGuard sguard(c10::hip::getStreamFromExternal((hipStream_t)citems[lo].stream, citems[lo].device));
    int64_t arena_floats = 0;  // low-rank layers: dW2 [c, kh, kw, d] into ONE zero-filled scratch, then the product's chain rule
    for (size_t i = lo; i < hi; ++i)
      if (citems[i].w2a.defined()) arena_floats += (int64_t)citems[i].c * citems[i].d * citems[i].geom[0] * citems[i].geom[1];
    Tensor arena;
    if (arena_floats > 0) arena = at::zeros({arena_floats}, citems[lo].g_rows.options().dtype(at::kFloat));
    std::vector<LycLokrConvWgradItem> raw(hi - lo);
    std::vector<LycLokrLrChainItem> chain;
    int64_t off = 0;
    for (size_t i = lo; i < hi; ++i) {
      const DeferredLokrConv& it = citems[i];
      float* dw2p = mfp(it.dw2p);
      if (it.w2a.defined()) {
        const int taps = it.geom[0] * it.geom[1];
        dw2p = arena.mutable_data_ptr<float>() + off;
        off += (int64_t)it.c * it.d * taps;
        chain.push_back(LycLokrLrChainItem{dw2p, cfp(it.w2a), cfp(it.w2b), mfp(it.d_w2a), mfp(it.d_w2b), it.c, it.d, (int)it.w2a.size(1), taps});
      }
      raw[i - lo] = LycLokrConvWgradItem{cptr(it.g_rows), cptr(it.x_rows), cfp(it.f1), mfp(it.dw1), dw2p, mptr(it.ws), it.B, it.H,
                                         it.W, it.dw1_blocks, it.a, it.b, it.c, it.d, it.geom[0], it.geom[1], it.geom[2], it.geom[3],
                                         it.geom[4], it.geom[5], it.geom[6], it.geom[7], it.alpha};
    }
    check_rc(lyc_lokr_conv_wgrad_group(raw.data(), (int)raw.size(), citems[lo].code, citems[lo].stream), "lyc_lokr_conv_wgrad_group");
    if (!chain.empty()) check_rc(lyc_lokr_lr_chain_group(chain.data(), (int)chain.size(), citems[lo].stream), "lyc_lokr_lr_chain_group");
    join_ambient(citems[lo].device, citems[lo].stream);
    lo = hi;
  }
  // one call per (device, stream, dtype) run of items, in arrival order
  for (size_t lo = 0; lo < items.size();) {
    size_t hi = lo + 1;
    while (hi < items.size() && items[hi].device == items[lo].device && items[hi].stream == items[lo].stream &&
           items[hi].code == items[lo].code)
      ++hi;
    const c10::DeviceGuard guard(c10::Device(c10::kCUDA, items[lo].device));
    const c10::hip::HIPStreamGuard sguard(c10::hip::getStreamFromExternal((hipStream_t)items[lo].stream, items[lo].device));
    // low-rank layers: dW2 [c, d] of each goes to a slice of ONE zero-filled scratch arena, then through the product's chain rule
    int64_t arena_floats = 0;
    for (size_t i = lo; i < hi; ++i) {
      if (items[i].w2a.defined()) arena_floats += (int64_t)items[i].c * items[i].d;
      if (items[i].w1a.defined()) arena_floats += (int64_t)items[i].a * items[i].b;
    }
    Tensor arena;
    if (arena_floats > 0) arena = at::zeros({arena_floats}, items[lo].g.options().dtype(at::kFloat));
    std::vector<LycLokrWgradItem> raw(hi - lo);
    std::vector<LycLokrLrChainItem> chain;
    int64_t off = 0;
    for (size_t i = lo; i < hi; ++i) {
      const DeferredLokr& it = items[i];
      float* dw2 = mfp(it.dw2);
      if (it.w2a.defined()) {
        dw2 = arena.mutable_data_ptr<float>() + off;
        off += (int64_t)it.c * it.d;
        chain.push_back(LycLokrLrChainItem{dw2, cfp(it.w2a), cfp(it.w2b), mfp(it.d_w2a), mfp(it.d_w2b), it.c, it.d, (int)it.w2a.size(1)});
      }
      float* dw1 = mfp(it.dw1);
      if (it.w1a.defined()) {
        dw1 = arena.mutable_data_ptr<float>() + off;
        off += (int64_t)it.a * it.b;
        chain.push_back(LycLokrLrChainItem{dw1, cfp(it.w1a), cfp(it.w1b), mfp(it.d_w1a), mfp(it.d_w1b), it.a, it.b, (int)it.w1a.size(1), 1});
      }
      raw[i - lo] = LycLokrWgradItem{cptr(it.g), cptr(it.x), cfp(it.f1), dw1, dw2, mptr(it.ws), it.M,
                                     it.a, it.b, it.c, it.d, it.alpha};
    }
    // the problem table of the one-launch-per-tile-class path lives in a caching-allocator block of the flush stream
    const int64_t tbytes = lyc_lokr_wgrad_table_bytes((int)raw.size());
    Tensor table = at::empty({tbytes}, items[lo].g.options().dtype(at::kByte));
    check_rc(lyc_lokr_wgrad_group_ws(raw.data(), (int)raw.size(), items[lo].code, table.mutable_data_ptr(), tbytes, items[lo].stream),
             "lyc_lokr_wgrad_group_ws");
    if (!chain.empty()) check_rc(lyc_lokr_lr_chain_group(chain.data(), (int)chain.size(), items[lo].stream), "lyc_lokr_lr_chain_group");
    join_ambient(items[lo].device, items[lo].stream);
    lo = hi;
  }
  for (size_t lo = 0; lo < litems.size();) {
    size_t hi = lo + 1;
    while (hi < litems.size() && litems[hi].device == litems[lo].device && litems[hi].stream == litems[lo].stream &&
           litems[hi].code == litems[lo].code)
      ++hi;
    std::vector<LycLoconWgradItem> raw(hi - lo);
    for (size_t i = lo; i < hi; ++i) {
      const DeferredLocon& it = litems[i];
      raw[i - lo] = LycLoconWgradItem{cptr(it.g), cptr(it.x), cfp(it.t), cfp(it.dt), mfp(it.dd), mfp(it.du), it.M,
                                      it.I, it.O, it.r, it.alpha};
    }
    const c10::DeviceGuard guard(c10::Device(c10::kCUDA, litems[lo].device));
    check_rc(lyc_locon_wgrad_group(raw.data(), (int)raw.size(), litems[lo].code, litems[lo].stream), "lyc_locon_wgrad_group");
    join_ambient(litems[lo].device, litems[lo].stream);
    lo = hi;
  }
  for (size_t lo = 0; lo < hitems.size();) {
    size_t hi = lo + 1;
    while (hi < hitems.size() && hitems[hi].device == hitems[lo].device && hitems[hi].stream == hitems[lo].stream &&
           hitems[hi].code == hitems[lo].code)
      ++hi;
    const c10::DeviceGuard guard(c10::Device(c10::kCUDA, hitems[lo].device));
    const c10::hip::HIPStreamGuard sguard(c10::hip::getStreamFromExternal((hipStream_t)hitems[lo].stream, hitems[lo].device));
    std::vector<LycLohaWgradItem> raw(hi - lo);
    std::vector<Tensor> gws(hi - lo);  // G = g^T x of every layer of the batch: [O, I] fp32 each (HBM is not the constraint here)
    for (size_t i = lo; i < hi; ++i) {
      const DeferredLoha& it = hitems[i];
      gws[i - lo] = at::empty({it.O, it.I}, it.x.options().dtype(at::kFloat));
      raw[i - lo] = LycLohaWgradItem{cptr(it.g), cptr(it.x), cfp(it.f[0]), cfp(it.f[1]), cfp(it.f[2]), cfp(it.f[3]),
                                     mfp(it.d[0]), mfp(it.d[1]), mfp(it.d[2]), mfp(it.d[3]), mfp(gws[i - lo]), it.M,
                                     it.I, it.O, it.r, it.alpha};
    }
    check_rc(lyc_loha_wgrad_group(raw.data(), (int)raw.size(), hitems[lo].code, hitems[lo].stream), "lyc_loha_wgrad_group");
    lo = hi;
  }
  for (size_t lo = 0; lo < hitems.size(); ++lo)
    if (lo + 1 == hitems.size() || hitems[lo + 1].stream != hitems[lo].stream) join_ambient(hitems[lo].device, hitems[lo].stream);
  // the gradients are enqueued: tell the DP sync (no lock held: this takes the GIL)
  std::vector<Tensor> ready;
  ready.reserve(4 * hitems.size() + 3 * items.size() + 2 * litems.size() + 3 * citems.size());
  for (const DeferredLoha& it : hitems)
    for (int i = 0; i < 4; ++i) ready.push_back(it.p[i]);
  for (const DeferredLokr& it : items) {
    if (it.w1a.defined()) {
      ready.push_back(it.w1a);
      ready.push_back(it.w1b);
    } else if (it.dw1.defined()) {
      ready.push_back(it.w1);
    }
    if (it.w2a.defined()) {
      ready.push_back(it.w2a);
      ready.push_back(it.w2b);
    } else {
      ready.push_back(it.w2);
    }
  }
  for (const DeferredLocon& it : litems) {
    if (it.dd.defined()) ready.push_back(it.down);
    if (it.du.defined()) ready.push_back(it.up);
  }
  for (const DeferredLokrConv& it : citems) {
    if (it.dw1.defined()) ready.push_back(it.w1);
    if (it.w2a.defined()) {
      ready.push_back(it.w2a);
      ready.push_back(it.w2b);
    } else {
      ready.push_back(it.w2);
    }
  }
  notify_many(ready);
}

// called from a backward node (the engine has a current graph task: final callbacks may be installed)
template <typename Item>
void park_deferred_in(std::vector<Item> DeferredLists::*list, Item&& item) {
  bool queue = false, full = false;
  // one final callback per backward pass (graph task) that parks something: a pass that died with an exception never runs
  // its callback, so "a callback is queued" must not be remembered across passes
  const int task = torch::autograd::get_current_graph_task_id();
  const c10::DeviceIndex device = item.device;
  {
    DeferredLists& L = lists_of(device);
    std::lock_guard<std::mutex> lock(L.mu);
    (L.*list).push_back(std::move(item));
    if (!L.callback_queued || L.queued_task != task) {
      L.callback_queued = queue = true;
      L.queued_task = task;
    }
    full = L.size() >= g_defer.flush_at.load();
  }
  if (queue) torch::autograd::Engine::get_default_engine().queue_callback([device]() { flush_deferred(device); });
  if (full) flush_deferred(device);
}
void park_deferred(DeferredLokr&& item) { park_deferred_in(&DeferredLists::lokr, std::move(item)); }
void park_deferred(DeferredLocon&& item) { park_deferred_in(&DeferredLists::locon, std::move(item)); }
void park_deferred(DeferredLoha&& item) { park_deferred_in(&DeferredLists::loha, std::move(item)); }
void park_deferred(DeferredLokrConv&& item) { park_deferred_in(&DeferredLists::lokr_conv, std::move(item)); }

// =====================================================================================================================
// Pre-packed LoKr operand planes (csrc/kron_conv.h): a cache keyed on the PARAMETER
// =====================================================================================================================
// w2 changes once per optimizer step; its hi / lo operand planes (forward and backward role, in the activation dtype) are kept
// per leaf tensor together with the parameter's version counter.  A layer call that finds its entry stale -- the first one after
// optimizer.step() -- refreshes EVERY stale entry of that device in one grouped launch (lyc_lokr_pack_group); the other layers
// of the step then just read.  Non-leaf factors (low-rank products, gated factors) are new tensors on every call: no planes,
// the kernels convert the fp32 tile themselves as before.  `refresh_planes(force)` is the explicit form for callers that
// replay captured graphs (the capture must contain the pack launch: bench.py).
struct PlaneEntry {
  c10::weak_intrusive_ptr<c10::TensorImpl> owner;
  Tensor planes[2];          // [bf16, f16]: fwd role bytes, then bwd role bytes
  int64_t version[2] = {-1, -1};
  int64_t epoch[2] = {-1, -1};  // PlaneCache::epoch the planes were packed in
  int c = 0, d = 0, taps = 0;
  int64_t sq = 0, sv = 0, st = 0;
  const float* w2 = nullptr;
  // low-rank pair (keyed on w2a): w2 == nullptr, planes packed from (w2a [c, rank], w2b [rank, d]); both version counters count
  c10::weak_intrusive_ptr<c10::TensorImpl> owner_b;
  const float* w2a = nullptr;
  const float* w2b = nullptr;
  int rank = 0;
  int64_t version_b[2] = {-1, -1};
  c10::DeviceIndex device = 0;
  explicit PlaneEntry(c10::weak_intrusive_ptr<c10::TensorImpl> o)
      : owner(std::move(o)), owner_b(c10::weak_intrusive_ptr<c10::TensorImpl>(c10::intrusive_ptr<c10::TensorImpl>())) {}
};
struct PlaneCache {
  std::mutex mu;
  std::unordered_map<const void*, PlaneEntry> map;
  std::atomic<bool> enabled{true};
  // the autograd version counter does not see writes through `p.data` (Prodigy, DAdaptation, 8-bit optimizers
  // with raw kernels, master-weight copies `p.data.copy_(master)`).  So a STEP BOUNDARY makes every entry stale as well: `epoch`
  // moves when `dirty` is found set by a forward-role lookup; `dirty` is set at the end of every backward pass that ran one of
  // the adapter ops (engine final callback) and by the global optimizer-step post hook (ops.py).  The first layer call of the
  // next forward pass then repacks every cached factor in one grouped launch -- what the version check already did for
  // optimizers that bump the counter.  Forward recomputation inside a backward pass (activation checkpointing) sees `dirty`
  // unset until that pass has ended.
  std::atomic<bool> dirty{false};
  int64_t epoch = 0;
  int queued_task = -1;  // graph task whose end-of-backward "dirty" callback is already queued
} g_planes;

// called from the backward nodes of the LoKr ops: ONE final callback per backward pass marks the planes dirty
void planes_mark_dirty_after_backward() {
  if (!g_planes.enabled) return;
  const int task = torch::autograd::get_current_graph_task_id();
  if (task < 0) return;
  {
    std::lock_guard<std::mutex> lk(g_planes.mu);
    if (g_planes.queued_task == task) return;
    g_planes.queued_task = task;
  }
  torch::autograd::Engine::get_default_engine().queue_callback([]() { g_planes.dirty = true; });
}
// forward-role lookups: a step boundary has passed -> new epoch (caller holds g_planes.mu)
void planes_new_epoch_if_dirty_locked(bool fwd_role) {
  if (fwd_role && g_planes.dirty.exchange(false)) ++g_planes.epoch;
}

// must hold g_planes.mu.  Repacks every entry of `device` whose parameter changed (or all with `force`), one grouped launch per
// 28 factors and dtype, on `stream`.
void refresh_planes_locked(c10::DeviceIndex device, void* stream, bool force) {
  for (int slot = 0; slot < 2; ++slot) {
    std::vector<LycLokrPackItem> items;
    std::vector<PlaneEntry*> who;
    std::vector<int64_t> vers, versb;
    for (auto it = g_planes.map.begin(); it != g_planes.map.end();) {
      PlaneEntry& e = it->second;
      auto owner = e.owner.lock();
      if (!owner) {  // the parameter is gone
        it = g_planes.map.erase(it);
        continue;
      }
      int64_t vb = -1;
      // the entry holds RAW pointers into the parameter's storage; `module.to(...)`, a `.data =` swap or
      // CPU offload may have replaced / freed it since this entry was last validated by its own layer call.  Check the live
      // owner before packing from the cached pointer; a mismatch drops the entry (the layer's next call starts over).
      auto live = [&](const c10::intrusive_ptr<c10::TensorImpl>& t, const void* cached) {
        return t->device_type() == c10::DeviceType::CUDA && t->device().index() == e.device && t->has_storage() &&
               t->dtype() == caffe2::TypeMeta::Make<float>() && t->data() == cached;
      };
      bool ok = true;
      if (e.rank > 0) {
        auto ob = e.owner_b.lock();
        if (!ob || !live(ob, e.w2b) || !live(owner, e.w2a)) ok = false;
        else vb = (int64_t)ob->version_counter().current_version();
      } else {
        ok = live(owner, e.w2) && owner->dim() >= 2 && owner->size(0) == e.c && owner->size(1) == e.d && owner->stride(0) == e.sq &&
             owner->stride(1) == e.sv;
      }
      if (!ok) {
        it = g_planes.map.erase(it);
        continue;
      }
      if (e.device == device && e.planes[slot].defined()) {
        const int64_t v = (int64_t)owner->version_counter().current_version();
        if (force || v != e.version[slot] || vb != e.version_b[slot] || e.epoch[slot] != g_planes.epoch) {
          char* base = static_cast<char*>(e.planes[slot].mutable_data_ptr());
          items.push_back(LycLokrPackItem{e.w2, e.sq, e.sv, e.st, e.c, e.d, e.taps, base, base + lyc_lokr_planes_bytes(e.c, e.d, e.taps, 0),
                                          e.w2a, e.w2b, e.rank});
          who.push_back(&e);
          vers.push_back(v);
          versb.push_back(vb);
        }
      }
      ++it;
    }
    if (items.empty()) continue;
    const c10::DeviceGuard guard(c10::Device(c10::kCUDA, device));
    if (items.size() > 28) {
      // round 6: ONE launch over all layers through a device table of the descriptors; the table is rewritten only when the list of
      // layers (pointers, shapes) differs from the one it was written for -- a training run writes it once
      struct PackTable {
        std::vector<LycLokrPackItem> last;
        Tensor table;
      };
      static std::unordered_map<int, PackTable> tables;  // (device, slot); guarded by g_planes.mu (the caller holds it)
      PackTable& pt = tables[(int)device * 2 + slot];
      auto same = [](const LycLokrPackItem& a, const LycLokrPackItem& b) {
        return a.w2 == b.w2 && a.sq == b.sq && a.sv == b.sv && a.st == b.st && a.c == b.c && a.d == b.d && a.taps == b.taps &&
               a.planes_fwd == b.planes_fwd && a.planes_bwd == b.planes_bwd && a.w2a == b.w2a && a.w2b == b.w2b && a.rank == b.rank;
      };
      bool valid = pt.table.defined() && pt.last.size() == items.size();
      for (size_t i = 0; valid && i < items.size(); ++i) valid = same(pt.last[i], items[i]);
      const int64_t tbytes = lyc_lokr_pack_table_bytes(items.data(), (int)items.size());
      if (!valid) {
        const c10::hip::HIPStreamGuard sguard(c10::hip::getStreamFromExternal((hipStream_t)stream, device));
        pt.table = at::empty({tbytes}, at::TensorOptions().device(c10::Device(c10::kCUDA, device)).dtype(at::kByte));
        pt.last = items;
      }
      check_rc(lyc_lokr_pack_group_ws(items.data(), (int)items.size(), slot == 0 ? LYC_BF16 : LYC_F16, pt.table.mutable_data_ptr(), tbytes,
                                      valid ? 1 : 0, stream), "lyc_lokr_pack_group_ws");
    } else {
      check_rc(lyc_lokr_pack_group(items.data(), (int)items.size(), slot == 0 ? LYC_BF16 : LYC_F16, stream), "lyc_lokr_pack_group");
    }
    for (size_t i = 0; i < who.size(); ++i) {
      who[i]->version[slot] = vers[i];
      who[i]->version_b[slot] = versb[i];
      who[i]->epoch[slot] = g_planes.epoch;
    }
  }
}

// planes of the leaf factor `w2` ([c, d] or [c, d, kh, kw]) for activations of `act`, or an undefined tensor
Tensor planes_for(const Tensor& w2, at::ScalarType act, void* stream, bool fwd_role = false) {
  if (!g_planes.enabled || !w2.defined() || !w2.is_cuda() || !w2.is_leaf() || w2.scalar_type() != at::kFloat) return Tensor();
  if (w2.is_inference()) return Tensor();  // no version counter: the kernels convert the fp32 tile themselves
  if (!fwd_role) planes_mark_dirty_after_backward();  // a backward pass is running: its end is a step boundary
  if (act != at::kBFloat16 && act != at::kHalf) return Tensor();
  const c10::DispatchKeySet ks = w2.key_set();
  if (ks.has(c10::DispatchKey::Python) || ks.has(c10::DispatchKey::Meta) || ks.has(c10::DispatchKey::Functionalize)) return Tensor();
  const int64_t c = w2.size(0), d = w2.size(1);
  int taps = 1;
  int64_t st = 0;
  if (w2.dim() == 4) {
    taps = (int)(w2.size(2) * w2.size(3));
    st = w2.stride(3);
    if (w2.size(2) > 1 && w2.stride(2) != w2.size(3) * w2.stride(3)) return Tensor();  // one tap stride: (i, j) -> i * kw + j
  } else if (w2.dim() != 2) {
    return Tensor();
  }
  if ((c % 8) != 0 || (d % 8) != 0 || c >= (1 << 20) || d >= (1 << 20)) return Tensor();
  const int slot = act == at::kBFloat16 ? 0 : 1;
  c10::TensorImpl* impl = w2.unsafeGetTensorImpl();
  std::lock_guard<std::mutex> lk(g_planes.mu);
  auto it = g_planes.map.find(impl);
  if (it != g_planes.map.end()) {
    auto owner = it->second.owner.lock();
    if (!owner || owner.get() != impl) {
      g_planes.map.erase(it);
      it = g_planes.map.end();
    }
  }
  if (it == g_planes.map.end())
    it = g_planes.map.emplace(impl, PlaneEntry(c10::weak_intrusive_ptr<c10::TensorImpl>(w2.getIntrusivePtr()))).first;
  PlaneEntry& e = it->second;
  const float* ptr = w2.const_data_ptr<float>();
  const bool same_view = e.w2 == ptr && e.c == c && e.d == d && e.taps == taps && e.sq == w2.stride(0) && e.sv == w2.stride(1) && e.st == st;
  if (!same_view) {  // first use, or the parameter's storage / layout changed (.data swap, .to(memory_format)): start over
    e.planes[0] = e.planes[1] = Tensor();
    e.version[0] = e.version[1] = -1;
    e.w2 = ptr; e.c = (int)c; e.d = (int)d; e.taps = taps; e.sq = w2.stride(0); e.sv = w2.stride(1); e.st = st;
    e.device = w2.device().index();
  }
  const int64_t v = (int64_t)w2._version();
  planes_new_epoch_if_dirty_locked(fwd_role);
  if (!e.planes[slot].defined()) {
    const int64_t nb = lyc_lokr_planes_bytes((int)c, (int)d, taps, 0) + lyc_lokr_planes_bytes((int)c, (int)d, taps, 1);
    e.planes[slot] = at::empty({nb}, w2.options().dtype(at::kByte));
    char* base = static_cast<char*>(e.planes[slot].mutable_data_ptr());
    check_rc(lyc_lokr_pack_w2(ptr, e.sq, e.sv, e.st, nullptr, 0, 0, nullptr, 0, 0, 0, 0, (int)c, (int)d, taps, base,
                              base + lyc_lokr_planes_bytes((int)c, (int)d, taps, 0), slot == 0 ? LYC_BF16 : LYC_F16, stream),
             "lyc_lokr_pack_w2");
    e.version[slot] = v;
    e.epoch[slot] = g_planes.epoch;
  } else if (e.version[slot] != v || e.epoch[slot] != g_planes.epoch) {
    refresh_planes_locked(e.device, stream, false);
    if (g_planes.map.find(impl) == g_planes.map.end()) return Tensor();  // (cannot happen for a just-validated entry; be safe)
  }
  return e.planes[slot];
}
// the same for a low-rank pair w2a [c, r], w2b [r, d] (both leaves, fp32, contiguous): planes of w2a @ w2b, formed in the pack kernel
Tensor planes_for_lr(const Tensor& w2a, const Tensor& w2b, at::ScalarType act, void* stream, int taps = 1, bool fwd_role = false) {
  if (w2a.defined() && w2b.defined() && (w2a.is_inference() || w2b.is_inference())) return Tensor();
  if (!fwd_role) planes_mark_dirty_after_backward();
  if (!g_planes.enabled || !w2a.is_cuda() || !w2a.is_leaf() || !w2b.is_leaf() || w2a.scalar_type() != at::kFloat ||
      w2b.scalar_type() != at::kFloat || !w2a.is_contiguous() || !w2b.is_contiguous() || w2a.dim() != 2 || w2b.dim() != 2)
    return Tensor();
  if (act != at::kBFloat16 && act != at::kHalf) return Tensor();
  if (taps < 1 || (w2b.size(1) % taps) != 0) return Tensor();
  const int64_t c = w2a.size(0), r = w2a.size(1), d = w2b.size(1) / taps;  // w2b [r, d * taps]: column v * taps + tap
  if ((c % 8) != 0 || (d % 8) != 0 || r < 1 || w2b.size(0) != r) return Tensor();
  const int slot = act == at::kBFloat16 ? 0 : 1;
  c10::TensorImpl* impl = w2a.unsafeGetTensorImpl();
  std::lock_guard<std::mutex> lk(g_planes.mu);
  auto it = g_planes.map.find(impl);
  if (it != g_planes.map.end()) {
    auto owner = it->second.owner.lock();
    if (!owner || owner.get() != impl) {
      g_planes.map.erase(it);
      it = g_planes.map.end();
    }
  }
  if (it == g_planes.map.end())
    it = g_planes.map.emplace(impl, PlaneEntry(c10::weak_intrusive_ptr<c10::TensorImpl>(w2a.getIntrusivePtr()))).first;
  PlaneEntry& e = it->second;
  const float *pa = w2a.const_data_ptr<float>(), *pb = w2b.const_data_ptr<float>();
  auto ob = e.owner_b.lock();
  const bool same_view = e.rank == r && e.w2a == pa && e.w2b == pb && e.c == c && e.d == d && e.taps == taps && ob &&
                         ob.get() == w2b.unsafeGetTensorImpl();
  if (!same_view) {
    e.planes[0] = e.planes[1] = Tensor();
    e.version[0] = e.version[1] = e.version_b[0] = e.version_b[1] = -1;
    e.w2 = nullptr; e.w2a = pa; e.w2b = pb; e.rank = (int)r; e.c = (int)c; e.d = (int)d; e.taps = taps; e.sq = e.sv = e.st = 0;
    e.owner_b = c10::weak_intrusive_ptr<c10::TensorImpl>(w2b.getIntrusivePtr());
    e.device = w2a.device().index();
  }
  const int64_t va = (int64_t)w2a._version(), vb = (int64_t)w2b._version();
  planes_new_epoch_if_dirty_locked(fwd_role);
  if (!e.planes[slot].defined()) {
    const int64_t nb = lyc_lokr_planes_bytes((int)c, (int)d, taps, 0) + lyc_lokr_planes_bytes((int)c, (int)d, taps, 1);
    e.planes[slot] = at::empty({nb}, w2a.options().dtype(at::kByte));
    char* base = static_cast<char*>(e.planes[slot].mutable_data_ptr());
    check_rc(lyc_lokr_pack_w2(nullptr, 0, 0, 0, pa, r, 1, pb, d * taps, taps, 1, (int)r, (int)c, (int)d, taps, base,
                              base + lyc_lokr_planes_bytes((int)c, (int)d, taps, 0), slot == 0 ? LYC_BF16 : LYC_F16, stream),
             "lyc_lokr_pack_w2(low rank)");
    e.version[slot] = va;
    e.version_b[slot] = vb;
    e.epoch[slot] = g_planes.epoch;
  } else if (e.version[slot] != va || e.version_b[slot] != vb || e.epoch[slot] != g_planes.epoch) {
    refresh_planes_locked(e.device, stream, false);
    if (g_planes.map.find(impl) == g_planes.map.end()) return Tensor();
  }
  return e.planes[slot];
}
// =====================================================================================================================
// LoHa operand planes (round 6): the same cache for dW = (w1a w1b) * (w2a w2b) * alpha, keyed on w1a
// =====================================================================================================================
// HadaWeight.forward (reference lycoris/functional/loha.py:10-16) rebuilds dW in every layer call; natively that was one 7 us launch per
// layer inside the forward pass (788 per SDXL step: 5.7 ms, profiles/r06_c13_loha_kernel_stats.csv).  The four factors are parameters:
// they change once per optimizer step.  Layers whose factors are contiguous fp32 leaves keep their 16-bit plane here (one per layer:
// 5 GB for the SDXL preset, of 288) and ALL stale planes of a device are rebuilt by one grouped launch per 48 layers
// (lyc_loha_rebuild_group) at the first layer call after a step boundary -- the staleness rules are the LoKr cache's (version counters
// of all four factors, the step epoch, `refresh_planes(force)` for captured steps).  The backward reads the same tensor.
struct LohaPlaneEntry {
  c10::weak_intrusive_ptr<c10::TensorImpl> owner[4];
  const float* ptr[4] = {nullptr, nullptr, nullptr, nullptr};
  Tensor plane[2];               // [bf16, f16]
  int64_t version[2][4];
  int64_t epoch[2] = {-1, -1};
  float alpha[2] = {0.f, 0.f};
  int O = 0, I = 0, r = 0;
  c10::DeviceIndex device = 0;
  LohaPlaneEntry()
      : owner{c10::weak_intrusive_ptr<c10::TensorImpl>(c10::intrusive_ptr<c10::TensorImpl>()),
              c10::weak_intrusive_ptr<c10::TensorImpl>(c10::intrusive_ptr<c10::TensorImpl>()),
              c10::weak_intrusive_ptr<c10::TensorImpl>(c10::intrusive_ptr<c10::TensorImpl>()),
              c10::weak_intrusive_ptr<c10::TensorImpl>(c10::intrusive_ptr<c10::TensorImpl>())} {
    for (auto& v : version)
      for (auto& x : v) x = -1;
  }
};
std::unordered_map<const void*, LohaPlaneEntry> g_loha_planes;  // guarded by g_planes.mu

// must hold g_planes.mu.  Rebuilds every stale plane of `device` (all with `force`), one launch per 48 layers and dtype.
void refresh_loha_planes_locked(c10::DeviceIndex device, void* stream, bool force) {
  for (int slot = 0; slot < 2; ++slot) {
    std::vector<LycLohaPlaneItem> items;
    std::vector<LohaPlaneEntry*> who;
    std::vector<std::array<int64_t, 4>> vers;
    for (auto it = g_loha_planes.begin(); it != g_loha_planes.end();) {
      LohaPlaneEntry& e = it->second;
      bool ok = true;
      std::array<int64_t, 4> v{};
      for (int k = 0; k < 4 && ok; ++k) {
        auto t = e.owner[k].lock();  // (raw pointers into the parameters' storage: validate the live owners, as the LoKr cache does)
        ok = t && t->device_type() == c10::DeviceType::CUDA && t->device().index() == e.device && t->has_storage() &&
             t->dtype() == caffe2::TypeMeta::Make<float>() && t->data() == e.ptr[k] && t->is_contiguous();
        if (ok) v[k] = (int64_t)t->version_counter().current_version();
      }
      if (!ok) {
        it = g_loha_planes.erase(it);
        continue;
      }
      if (e.device == device && e.plane[slot].defined()) {
        bool stale = force || e.epoch[slot] != g_planes.epoch;
        for (int k = 0; k < 4; ++k) stale = stale || v[k] != e.version[slot][k];
        if (stale) {
          items.push_back(LycLohaPlaneItem{e.ptr[0], e.ptr[1], e.ptr[2], e.ptr[3], e.plane[slot].mutable_data_ptr(), e.O, e.I, e.r, e.alpha[slot]});
          who.push_back(&e);
          vers.push_back(v);
        }
      }
      ++it;
    }
    if (items.empty()) continue;
    const c10::DeviceGuard guard(c10::Device(c10::kCUDA, device));
    check_rc(lyc_loha_rebuild_group(items.data(), (int)items.size(), slot == 0 ? LYC_BF16 : LYC_F16, stream), "lyc_loha_rebuild_group");
    for (size_t i = 0; i < who.size(); ++i) {
      for (int k = 0; k < 4; ++k) who[i]->version[slot][k] = vers[i][k];
      who[i]->epoch[slot] = g_planes.epoch;
    }
  }
}

// the cached operand plane of a layer whose four factors are contiguous fp32 leaf parameters, for activations of `act`; or undefined
Tensor loha_plane_for(const Tensor& w1a, const Tensor& w1b, const Tensor& w2a, const Tensor& w2b, double alpha, at::ScalarType act,
                      void* stream, bool fwd_role) {
  if (!g_planes.enabled || (act != at::kBFloat16 && act != at::kHalf)) return Tensor();
  const Tensor* f[4] = {&w1a, &w1b, &w2a, &w2b};
  for (const Tensor* t : f) {
    if (!t->defined() || !t->is_cuda() || !t->is_leaf() || t->scalar_type() != at::kFloat || !t->is_contiguous() || t->dim() != 2 ||
        t->is_inference() || t->device() != w1a.device())
      return Tensor();
    const c10::DispatchKeySet ks = t->key_set();
    if (ks.has(c10::DispatchKey::Python) || ks.has(c10::DispatchKey::Meta) || ks.has(c10::DispatchKey::Functionalize)) return Tensor();
  }
  const int64_t O = w1a.size(0), r = w1a.size(1), I = w1b.size(1);
  if (w2a.size(0) != O || w2a.size(1) != r || w1b.size(0) != r || w2b.size(0) != r || w2b.size(1) != I || O >= (1 << 30) || I >= (1 << 30))
    return Tensor();
  const int slot = act == at::kBFloat16 ? 0 : 1;
  const int code = slot == 0 ? LYC_BF16 : LYC_F16;
  const float* ptr[4];
  for (int k = 0; k < 4; ++k) ptr[k] = f[k]->const_data_ptr<float>();
  if (!lyc_loha_plane_cacheable(ptr[0], ptr[1], ptr[2], ptr[3], (int)I, (int)O, (int)r, code)) return Tensor();
  c10::TensorImpl* impl = w1a.unsafeGetTensorImpl();
  std::lock_guard<std::mutex> lk(g_planes.mu);
  auto it = g_loha_planes.find(impl);
  if (it == g_loha_planes.end()) it = g_loha_planes.emplace(impl, LohaPlaneEntry()).first;
  LohaPlaneEntry& e = it->second;
  bool same_view = e.O == O && e.I == I && e.r == r;
  for (int k = 0; k < 4 && same_view; ++k) {
    auto o = e.owner[k].lock();
    same_view = o && o.get() == f[k]->unsafeGetTensorImpl() && e.ptr[k] == ptr[k];
  }
  if (!same_view) {  // first use, or a parameter's storage changed: start over
    e.plane[0] = e.plane[1] = Tensor();
    for (int k = 0; k < 4; ++k) {
      e.owner[k] = c10::weak_intrusive_ptr<c10::TensorImpl>(f[k]->getIntrusivePtr());
      e.ptr[k] = ptr[k];
    }
    e.O = (int)O; e.I = (int)I; e.r = (int)r; e.device = w1a.device().index();
  }
  planes_new_epoch_if_dirty_locked(fwd_role);
  int64_t v[4];
  for (int k = 0; k < 4; ++k) v[k] = (int64_t)f[k]->_version();
  auto build_one = [&]() {
    LycLohaPlaneItem item{ptr[0], ptr[1], ptr[2], ptr[3], e.plane[slot].mutable_data_ptr(), (int)O, (int)I, (int)r, (float)alpha};
    check_rc(lyc_loha_rebuild_group(&item, 1, code, stream), "lyc_loha_rebuild_group");
    for (int k = 0; k < 4; ++k) e.version[slot][k] = v[k];
    e.epoch[slot] = g_planes.epoch;
    e.alpha[slot] = (float)alpha;
  };
  if (!e.plane[slot].defined()) {
    e.plane[slot] = at::empty({lyc_loha_workspace_bytes((int)O, (int)I, code)}, w1a.options().dtype(at::kByte));
    build_one();
  } else if (e.alpha[slot] != (float)alpha) {  // another multiplier / alpha: this layer alone, now
    build_one();
  } else {
    bool stale = e.epoch[slot] != g_planes.epoch;
    for (int k = 0; k < 4; ++k) stale = stale || e.version[slot][k] != v[k];
    if (stale) {
      refresh_loha_planes_locked(e.device, stream, false);
      if (g_loha_planes.find(impl) == g_loha_planes.end()) return Tensor();
    }
  }
  return e.plane[slot];
}

const void* planes_bwd_ptr(const Tensor& planes, int64_t c, int64_t d, int taps) {
  return static_cast<const char*>(planes.const_data_ptr()) + lyc_lokr_planes_bytes((int)c, (int)d, taps, 0);
}

// =====================================================================================================================
// LoKr on nn.Linear
// =====================================================================================================================
Tensor lokr_linear_fwd(const Tensor& x, const Tensor& w1, const Tensor& w2, double alpha, const c10::optional<Tensor>& base) {
  require_device(x, "input");
  const c10::DeviceGuard guard(x.device());
  // (w2 [c, d, 1, 1]: the factor of a 1x1 convolution, contiguous -- [c, d] in memory; ops.lokr_conv2d hands the leaf over unreshaped)
  TORCH_CHECK(w1.dim() == 2 && (w2.dim() == 2 || (w2.dim() == 4 && w2.size(2) == 1 && w2.size(3) == 1 && w2.is_contiguous())),
              "lokr_linear: w1 [a, b], w2 [c, d] (or the contiguous [c, d, 1, 1] of a 1x1 convolution)");
  const int64_t a = w1.size(0), b = w1.size(1), c = w2.size(0), d = w2.size(1);
  TORCH_CHECK(x.size(-1) == b * d, "adapter expects ", b * d, " input features, got ", x.sizes());
  Tensor rows = rows_of(x, b * d), f1 = f32c(w1);
  auto oshape = x.sizes().vec();
  oshape.back() = a * c;
  Tensor y = at::empty({rows.size(0), a * c}, x.options());
  Tensor bs;
  if (base.has_value() && base->defined()) {
    TORCH_CHECK(base->scalar_type() == x.scalar_type() && base->numel() == y.numel() && base->is_contiguous(),
                "lokr_linear: `base` must be the frozen layer's contiguous output in the activation dtype");
    bs = *base;
  }
  const int code = dtype_code(x.scalar_type());
  Tensor pl;
  if (lyc_lokr_linear_planes_ok(rows.size(0), (int)a, (int)b, (int)c, (int)d, code) && (reinterpret_cast<uintptr_t>(cptr(rows)) & 15u) == 0)
    pl = planes_for(w2, x.scalar_type(), stream_of(x), /*fwd_role=*/true);
  if (pl.defined()) {
    check_rc(lyc_lokr_linear_fwd_planes(cptr(rows), cfp(f1), cptr(pl), cptr(bs), mptr(y), rows.size(0), (int)a, (int)b, (int)c, (int)d,
                                        (float)alpha, code, stream_of(x)), "lyc_lokr_linear_fwd_planes");
  } else {
    Tensor f2 = f32c(w2);
    check_rc(lyc_lokr_linear_fwd(cptr(rows), cfp(f1), cfp(f2), cptr(bs), mptr(y), rows.size(0), (int)a, (int)b, (int)c, (int)d,
                                 (float)alpha, code, stream_of(x)), "lyc_lokr_linear_fwd");
  }
  return y.view(oshape);
}

// dx (or undefined), and the factor gradients accumulated into dw1 / dw2 when those are defined
Tensor lokr_linear_bwd_into(const Tensor& g, const Tensor& x, const Tensor& w1, const Tensor& w2, double alpha, bool need_dx,
                            const Tensor& dw1, const Tensor& dw2, bool f32_rows = false) {
  const c10::DeviceGuard guard(x.device());
  const int64_t a = w1.size(0), b = w1.size(1), c = w2.size(0), d = w2.size(1);
  Tensor rows = rows_of(x, b * d), g2 = rows_of(g, a * c), f1 = f32c(w1);
  const int code0 = dtype_code(x.scalar_type()), code = code0 | (f32_rows ? LYC_F32_ROWS : 0);  // f32_rows: dx in fp32 (the Conv2d lowering's col2im rounds once)
  const bool want_dx = need_dx || dw1.defined();  // the w1 gradient shares the pass over g that produces dx
  Tensor dx, ws;
  if (want_dx) dx = at::empty(rows.sizes(), f32_rows ? x.options().dtype(at::kFloat) : x.options());
  if (dw1.defined()) {
    const int64_t nbytes = lyc_lokr_bwd_workspace_bytes(rows.size(0), (int)a, (int)b, (int)c, (int)d, code0);
    if (nbytes > 0) ws = at::empty({nbytes}, x.options().dtype(at::kByte));
  }
  Tensor pl;
  if (want_dx && !f32_rows && lyc_lokr_linear_planes_ok(rows.size(0), (int)a, (int)b, (int)c, (int)d, code0) &&
      (reinterpret_cast<uintptr_t>(cptr(g2)) & 15u) == 0)
    pl = planes_for(w2, x.scalar_type(), stream_of(x));
  if (pl.defined()) {
    check_rc(lyc_lokr_linear_bwd_planes(cptr(g2), cptr(rows), cfp(f1), planes_bwd_ptr(pl, c, d, 1), mptr(dx), mfp(dw1), mfp(dw2), mptr(ws),
                                        rows.size(0), (int)a, (int)b, (int)c, (int)d, (float)alpha, code, stream_of(x)),
             "lyc_lokr_linear_bwd_planes");
  } else {
    Tensor f2 = f32c(w2);
    check_rc(lyc_lokr_linear_bwd(cptr(g2), cptr(rows), cfp(f1), cfp(f2), mptr(dx), mfp(dw1), mfp(dw2), mptr(ws), rows.size(0),
                                 (int)a, (int)b, (int)c, (int)d, (float)alpha, code, stream_of(x)), "lyc_lokr_linear_bwd");
  }
  return need_dx ? shaped_like(dx, x) : Tensor();
}

// dx now, dw1 / dw2 later (park_deferred): returns false when the layer is not on the grouped fast path
bool lokr_linear_bwd_deferred(const Tensor& g, const Tensor& x, const Tensor& w1, const Tensor& w2, double alpha, bool need_dx,
                              const Tensor& dw1, const Tensor& dw2, Tensor& dx_out) {
  const c10::DeviceGuard guard(x.device());
  const int64_t a = w1.size(0), b = w1.size(1), c = w2.size(0), d = w2.size(1);
  Tensor rows = rows_of(x, b * d), g2 = rows_of(g, a * c);
  const int code = dtype_code(x.scalar_type());
  if (!lyc_lokr_wgrad_deferrable(cptr(g2), cptr(rows), rows.size(0), (int)a, (int)b, (int)c, (int)d, code)) return false;
  Tensor f1 = f32c(w1), dx, ws;
  const bool want_dx = need_dx || dw1.defined();
  if (want_dx) {
    dx = at::empty(rows.sizes(), x.options());
    if (dw1.defined()) {
      const int64_t nbytes = lyc_lokr_bwd_workspace_bytes(rows.size(0), (int)a, (int)b, (int)c, (int)d, code);
      TORCH_CHECK(nbytes > 0, "lycoris_amd: deferrable layer without a dw1 workspace");
      ws = at::empty({nbytes}, x.options().dtype(at::kByte));
    }
    Tensor pl = planes_for(w2, x.scalar_type(), stream_of(x));  // deferrable layers are on the 16-bit fast path
    if (pl.defined()) {
      check_rc(lyc_lokr_linear_bwd_planes(cptr(g2), cptr(rows), cfp(f1), planes_bwd_ptr(pl, c, d, 1), mptr(dx), mfp(dw1), nullptr,
                                          mptr(ws), rows.size(0), (int)a, (int)b, (int)c, (int)d, (float)alpha, code | LYC_DEFER_WGRAD,
                                          stream_of(x)), "lyc_lokr_linear_bwd_planes(dx)");
    } else {
      Tensor f2 = f32c(w2);
      check_rc(lyc_lokr_linear_bwd(cptr(g2), cptr(rows), cfp(f1), cfp(f2), mptr(dx), mfp(dw1), nullptr, mptr(ws), rows.size(0),
                                   (int)a, (int)b, (int)c, (int)d, (float)alpha, code | LYC_DEFER_WGRAD, stream_of(x)),
               "lyc_lokr_linear_bwd(dx)");
    }
  }
  park_deferred(DeferredLokr{g2, rows, f1, w1, w2, dw1, dw2, ws, rows.size(0), (int)a, (int)b, (int)c, (int)d, code,
                             (float)alpha, stream_of(x), x.device().index()});
  dx_out = need_dx ? shaped_like(dx, x) : Tensor();
  return true;
}

std::tuple<Tensor, Tensor, Tensor> lokr_linear_bwd(const Tensor& g, const Tensor& x, const Tensor& w1, const Tensor& w2,
                                                   double alpha, bool need_dx, bool need_dw1, bool need_dw2) {
  Tensor dw1 = need_dw1 ? at::zeros(w1.sizes(), w1.options().dtype(at::kFloat)) : Tensor();
  Tensor dw2 = need_dw2 ? at::zeros(w2.sizes(), w2.options().dtype(at::kFloat)) : Tensor();
  Tensor dx = lokr_linear_bwd_into(g, x, w1, w2, alpha, need_dx, dw1, dw2);
  return {dx.defined() ? dx : at::empty({0}, x.options()), need_dw1 ? dw1.to(w1.scalar_type()) : at::empty({0}, w1.options()),
          need_dw2 ? dw2.to(w2.scalar_type()) : at::empty({0}, w2.options())};
}

struct LokrLinearFn : public torch::autograd::Function<LokrLinearFn> {
  static Tensor forward(AutogradContext* ctx, const Tensor& x, const Tensor& w1, const Tensor& w2, double alpha,
                        const c10::optional<Tensor>& base) {
    at::AutoDispatchBelowADInplaceOrView guard;
    static auto op = c10::Dispatcher::singleton().findSchemaOrThrow("lycoris_amd::lokr_linear", "")
                         .typed<Tensor(const Tensor&, const Tensor&, const Tensor&, double, const c10::optional<Tensor>&)>();
    // eager device tensors: straight to the kernel launch (the second trip through the dispatcher is ~1 us of host time per layer)
    Tensor y = eager_cuda(x) ? lokr_linear_fwd(x, w1, w2, alpha, base) : op.call(x, w1, w2, alpha, base);
    expect(w1, x);
    expect(w2, x);
    save_vars(ctx, {x, w1, w2});
    ctx->saved_data["alpha"] = alpha;
    ctx->saved_data["has_base"] = base.has_value() && base->defined();
    return y;
  }
  static variable_list backward(AutogradContext* ctx, variable_list grads) {
    planes_mark_dirty_after_backward();  // the end of this backward pass is a step boundary for the plane cache (every LoKr node, on every path)
    auto saved = saved_vars(ctx);
    const Tensor &x = saved[0], &w1 = saved[1], &w2 = saved[2];
    const double alpha = ctx->saved_data["alpha"].toDouble();
    const bool nx = ctx->needs_input_grad(0), n1 = ctx->needs_input_grad(1), n2 = ctx->needs_input_grad(2);
    // needs_input_grad indexes the VARIABLE inputs (x, w1, w2[, base]); the float `alpha` has no edge
    const bool nb = ctx->saved_data["has_base"].toBool() && ctx->needs_input_grad(3);
    Tensor g = grads[0];
    if (eager_cuda(g) && eager_cuda(x)) {  // eager: accumulate straight into .grad where possible
      GradTarget t1 = grad_target(w1, n1), t2 = grad_target(w2, n2);
      if (g_defer.enabled && t2.buf.defined() && !t2.hand_back && !(t1.buf.defined() && t1.hand_back)) {
        Tensor dx;
        if (lokr_linear_bwd_deferred(g, x, w1, w2, alpha, nx, t1.buf, t2.buf, dx))
          return {dx, Tensor(), Tensor(), Tensor(), nb ? g : Tensor()};
      }
      Tensor dx = lokr_linear_bwd_into(g, x, w1, w2, alpha, nx, t1.buf, t2.buf);
      return {dx, finish_grad(w1, t1), finish_grad(w2, t2), Tensor(), nb ? g : Tensor()};  // d(base + delta)/d base = 1
    }
    static auto op = c10::Dispatcher::singleton().findSchemaOrThrow("lycoris_amd::_lokr_linear_backward", "")
                         .typed<std::tuple<Tensor, Tensor, Tensor>(const Tensor&, const Tensor&, const Tensor&, const Tensor&,
                                                                   double, bool, bool, bool)>();
    auto [dx, d1, d2] = op.call(g, x, w1, w2, alpha, nx, n1, n2);
    return {nx ? dx : Tensor(), n1 ? d1 : Tensor(), n2 ? d2 : Tensor(), Tensor(), nb ? g : Tensor()};
  }
};

Tensor lokr_linear_autograd(const Tensor& x, const Tensor& w1, const Tensor& w2, double alpha, const c10::optional<Tensor>& base) {
  const GradAtApply ga_;
  return LokrLinearFn::apply(amp(x), w1, w2, alpha, base);
}

Tensor lokr_linear_meta(const Tensor& x, const Tensor& w1, const Tensor& w2, double alpha, const c10::optional<Tensor>& base) {
  auto oshape = x.sym_sizes().vec();
  oshape.back() = w1.sym_size(0) * w2.sym_size(0);
  return x.new_empty_symint(oshape);
}
std::tuple<Tensor, Tensor, Tensor> lokr_linear_bwd_meta(const Tensor& g, const Tensor& x, const Tensor& w1, const Tensor& w2,
                                                        double alpha, bool need_dx, bool need_dw1, bool need_dw2) {
  return {need_dx ? at::empty_like(x) : x.new_empty({0}), need_dw1 ? at::empty_like(w1) : w1.new_empty({0}),
          need_dw2 ? at::empty_like(w2) : w2.new_empty({0})};
}

// =====================================================================================================================
// Sibling projections of ONE input in one launch (round 5; VERDICT r4 #4: "wire the sibling groups into the modules")
// =====================================================================================================================
// to_q / to_k / to_v of a self-attention block and to_k / to_v of a cross-attention read the same tensor; the reference calls one
// LokrModule.forward per projection (modules/lokr.py:543-566).  `lokr_linear_group(x, factors, alphas, bases)` is those n forwards as
// ONE dispatcher call, ONE forward launch (lyc_lokr_linear_fwd_group, with the fused `base + delta` epilogue when bases are given),
// ONE autograd node with n outputs and ONE backward launch that stores the SUM of the n dx results (lyc_lokr_linear_bwd_group_sum);
// the weight gradients are parked per problem exactly like a single layer's.  Forward results are bit-identical to n lokr_linear calls
// (tests/test_gpu_siblings.py).  Anything not on the packed-plane fast path runs the problems one by one through the single-layer
// functions -- same numbers, n launches.
//   F = 2 (lokr_linear_group)   : factors = [w1_0, w2_0, w1_1, w2_1, ...]            full-matrix w2 [c, d]
//   F = 3 (lokr_linear_lr_group): factors = [w1_0, w2a_0, w2b_0, w1_1, ...]          low-rank w2 = w2a [c, r] @ w2b [r, d] (lokr.py:131-136):
//                                 planes packed from the pair, dW2 through the grouped chain rule, as lokr_linear_lr
//   bases = [] or n tensors; all problems share (a, b, c, d)
Tensor lokr_linear_lr_fwd(const Tensor& x, const Tensor& w1, const Tensor& w2a, const Tensor& w2b, double alpha, const c10::optional<Tensor>& base);
std::tuple<Tensor, Tensor, Tensor, Tensor> lokr_linear_lr_bwd_one(const Tensor& g, const Tensor& x, const Tensor& w1, const Tensor& w2a,
                                                                  const Tensor& w2b, double alpha, bool nx, bool n1, bool na, bool nb2);

std::vector<Tensor> lokr_group_fwd_impl(const Tensor& x, at::TensorList factors, at::ArrayRef<double> alphas, at::TensorList bases, const size_t F) {
  require_device(x, "input");
  const c10::DeviceGuard guard(x.device());
  const size_t n = alphas.size();
  TORCH_CHECK(n >= 1 && factors.size() == F * n && (bases.empty() || bases.size() == n), "lokr_linear_group: n alphas, ", F, "n factors, 0 or n bases");
  const Tensor& w10 = factors[0];
  TORCH_CHECK(w10.dim() == 2 && factors[1].dim() == 2 && factors[F - 1].dim() == 2, "lokr_linear_group: 2-D factors");
  const int64_t a = w10.size(0), b = w10.size(1), c = factors[1].size(0), d = factors[F - 1].size(1);
  for (size_t i = 0; i < n; ++i)
    for (size_t j = 0; j < F; ++j)
      TORCH_CHECK(factors[F * i + j].sizes() == factors[j].sizes(), "lokr_linear_group: the problems of a group share their factor shapes");
  TORCH_CHECK(x.size(-1) == b * d, "adapter expects ", b * d, " input features, got ", x.sizes());
  Tensor rows = rows_of(x, b * d);
  auto oshape = x.sizes().vec();
  oshape.back() = a * c;
  const int code = dtype_code(x.scalar_type());
  const int64_t M = rows.size(0);
  std::vector<Tensor> ys(n), pls(n), f1s(n);
  bool fast = n >= 2 && lyc_lokr_linear_planes_ok(M, (int)a, (int)b, (int)c, (int)d, code) && (reinterpret_cast<uintptr_t>(cptr(rows)) & 15u) == 0;
  for (size_t i = 0; i < n && fast; ++i) {
    if (!bases.empty()) {
      const Tensor& bs = bases[i];
      fast = bs.defined() && bs.scalar_type() == x.scalar_type() && bs.numel() == M * a * c && bs.is_contiguous() &&
             (reinterpret_cast<uintptr_t>(cptr(bs)) & 15u) == 0;
      if (!fast) break;
    }
    pls[i] = F == 2 ? planes_for(factors[2 * i + 1], x.scalar_type(), stream_of(x), /*fwd_role=*/true)
                    : planes_for_lr(factors[3 * i + 1], factors[3 * i + 2], x.scalar_type(), stream_of(x), 1, /*fwd_role=*/true);
    fast = pls[i].defined();
  }
  if (!fast) {  // one by one: the single-layer forward (which picks its own kernel); `base` added on this side when it cannot be fused
    for (size_t i = 0; i < n; ++i) {
      c10::optional<Tensor> bs = bases.empty() ? c10::nullopt : c10::optional<Tensor>(bases[i]);
      ys[i] = F == 2 ? lokr_linear_fwd(x, factors[2 * i], factors[2 * i + 1], alphas[i], bs)
                     : lokr_linear_lr_fwd(x, factors[3 * i], factors[3 * i + 1], factors[3 * i + 2], alphas[i], bs);
    }
    return ys;
  }
  std::vector<LycLokrLinearGroupItem> items(n);
  for (size_t i = 0; i < n; ++i) {
    f1s[i] = f32c(factors[F * i]);
    ys[i] = at::empty({M, a * c}, x.options());
    items[i] = LycLokrLinearGroupItem{cptr(rows), cfp(f1s[i]), cptr(pls[i]), bases.empty() ? nullptr : cptr(bases[i]), mptr(ys[i]), nullptr, M,
                                      (float)alphas[i]};
  }
  check_rc(lyc_lokr_linear_fwd_group(items.data(), (int)n, (int)a, (int)b, (int)c, (int)d, code, stream_of(x)), "lyc_lokr_linear_fwd_group");
  for (size_t i = 0; i < n; ++i) ys[i] = ys[i].view(oshape);
  return ys;
}
std::vector<Tensor> lokr_linear_group_fwd(const Tensor& x, at::TensorList factors, at::ArrayRef<double> alphas, at::TensorList bases) {
  return lokr_group_fwd_impl(x, factors, alphas, bases, 2);
}
std::vector<Tensor> lokr_linear_lr_group_fwd(const Tensor& x, at::TensorList factors, at::ArrayRef<double> alphas, at::TensorList bases) {
  return lokr_group_fwd_impl(x, factors, alphas, bases, 3);
}

struct LokrLinearGroupFn : public torch::autograd::Function<LokrLinearGroupFn> {
  // vars = [x, F factors per problem ..., base_0 ... base_{n-1} (optional)]
  // own != 0 (round 6, lokr_adapted_linear): vars = [x, F factors per problem ..., W_0 ... W_{n-1}, bias_0 ... bias_{n-1} (undefined: none)]
  // -- the node OWNS the frozen layers: base_i = x W_i^T + bias_i is formed here (library GEMM), and the backward adds g_i W_i into the
  // adapter's dx with the GEMM's accumulate epilogue (dx.addmm_): x has ONE consumer, the engine's `dx_base + dx_adapter` pass is gone.
  // (`frozen` = [W_0 ... W_{n-1}, bias_0 ... bias_{n-1}]: a std::vector<Tensor> argument is not an autograd input -- they have no gradient)
  static variable_list forward(AutogradContext* ctx, at::TensorList vars, std::vector<double> alphas, int64_t F_, std::vector<Tensor> frozen = {}) {
    at::AutoDispatchBelowADInplaceOrView guard;
    const size_t n = alphas.size(), F = (size_t)F_;
    const bool own = !frozen.empty();
    const bool has_base = !own && vars.size() == 1 + (F + 1) * n;
    TORCH_CHECK(own ? (vars.size() == 1 + F * n && frozen.size() == 2 * n) : (vars.size() == 1 + F * n || has_base), "lokr_linear_group: bad argument list");
    const Tensor& x = vars[0];
    at::TensorList factors = vars.slice(1, F * n);
    std::vector<Tensor> own_bases;
    if (own) {
      TORCH_CHECK(eager_cuda(x), "lycoris_amd::lokr_adapted_linear is an eager op on device tensors");
      for (size_t i = 0; i < n; ++i) {
        const Tensor &W = frozen[i], &bias = frozen[n + i];
        TORCH_CHECK(W.dim() == 2 && W.scalar_type() == x.scalar_type() && !W.requires_grad() && !(bias.defined() && bias.requires_grad()),
                    "lokr_adapted_linear: frozen 2-D weights in the activation dtype");
        own_bases.push_back(at::linear(x, W, bias.defined() ? c10::optional<Tensor>(bias) : c10::nullopt).contiguous());
      }
    }
    at::TensorList bases = own ? at::TensorList(own_bases) : (has_base ? vars.slice(1 + F * n, n) : at::TensorList());
    std::vector<Tensor> ys;
    if (eager_cuda(x)) {
      ys = lokr_group_fwd_impl(x, factors, alphas, bases, F);
    } else {
      TORCH_CHECK(F == 2, "lycoris_amd::lokr_linear_lr_group is an eager op (trace the products through lokr_linear_group instead)");
      static auto op = c10::Dispatcher::singleton().findSchemaOrThrow("lycoris_amd::lokr_linear_group", "")
                           .typed<std::vector<Tensor>(const Tensor&, at::TensorList, at::ArrayRef<double>, at::TensorList)>();
      ys = op.call(x, factors, alphas, bases);
    }
    for (size_t i = 0; i < F * n; ++i) expect(vars[1 + i], x);
    ctx->saved_data["alphas"] = alphas;
    ctx->saved_data["has_base"] = has_base;
    ctx->saved_data["n"] = (int64_t)n;
    ctx->saved_data["F"] = F_;
    ctx->saved_data["own"] = own;
    if (own) ctx->saved_data["lyc_frozen_w"] = c10::List<Tensor>(std::vector<Tensor>(frozen.begin(), frozen.begin() + n));
    // (saved by reference in saved_data where they are leaves: save_vars' identity rule, for up to 1 + 3 * 4 tensors)
    variable_list keep(vars.begin(), vars.begin() + 1 + F * n);
    for (size_t i = 1; i < keep.size(); ++i)
      if (keep[i].defined() && keep[i].is_leaf() && keep[i].requires_grad()) ctx->saved_data["lyc_gleaf" + std::to_string(i)] = keep[i];
    ctx->save_for_backward(std::move(keep));
    return ys;
  }
  static variable_list backward(AutogradContext* ctx, variable_list grads) {
    planes_mark_dirty_after_backward();
    variable_list s = ctx->get_saved_variables();
    for (size_t i = 1; i < s.size(); ++i) {  // the parameters themselves, not the copies saved-tensor hooks hand back (see save_vars)
      auto it = ctx->saved_data.find("lyc_gleaf" + std::to_string(i));
      if (it == ctx->saved_data.end() || !it->second.isTensor()) continue;
      const Tensor& p = it->second.toTensor();
      if (p.defined() && s[i].defined() && !s[i].is_same(p) && s[i].sizes() == p.sizes()) s[i] = p;
    }
    const size_t n = (size_t)ctx->saved_data["n"].toInt(), F = (size_t)ctx->saved_data["F"].toInt();
    const std::vector<double> alphas = ctx->saved_data["alphas"].toDoubleVector();
    const bool has_base = ctx->saved_data["has_base"].toBool();
    const bool own = ctx->saved_data["own"].toBool();
    const Tensor& x = s[0];
    const bool nx = ctx->needs_input_grad(0);
    variable_list out(1 + (F + (has_base ? 1 : 0)) * n);
    const c10::DeviceGuard guard(x.device());
    const int64_t a = s[1].size(0), b = s[1].size(1), c = s[2].size(0), d = s[F].size(1);
    const int code = x.defined() && x.is_cuda() ? dtype_code(x.scalar_type()) : 0;
    auto W1 = [&](size_t i) -> const Tensor& { return s[1 + F * i]; };
    auto W2 = [&](size_t i, size_t j) -> const Tensor& { return s[2 + F * i + j]; };  // F = 2: j = 0 (w2); F = 3: j = 0 / 1 (w2a / w2b)
    // ---- the grouped fast path: every problem deferrable (fused accumulation into .grad, 16-bit planes), every grad defined -------
    bool fast = n >= 2 && g_defer.enabled && eager_cuda(x);
    std::vector<GradTarget> t1(n), t2(n), t3(n);
    std::vector<Tensor> g2(n), pl(n);
    Tensor rows;
    if (fast) {
      rows = rows_of(x, b * d);
      for (size_t i = 0; i < n && fast; ++i) {
        fast = grads[i].defined() && eager_cuda(grads[i]);
        if (!fast) break;
        t1[i] = grad_target(W1(i), ctx->needs_input_grad(1 + F * i));
        fast = t1[i].buf.defined() && !t1[i].hand_back;
        if (!fast) break;
        if (F == 2) {
          t2[i] = grad_target(W2(i, 0), ctx->needs_input_grad(2 + 2 * i));
          fast = t2[i].buf.defined() && !t2[i].hand_back;
        } else {  // both halves of the product are wanted together (one dW2 scratch, one chain-rule item); fp32 contiguous leaves
          const bool want = ctx->needs_input_grad(2 + 3 * i) && ctx->needs_input_grad(3 + 3 * i);
          t2[i] = grad_target(W2(i, 0), want);
          t3[i] = grad_target(W2(i, 1), want);
          fast = want && t2[i].buf.defined() && !t2[i].hand_back && t3[i].buf.defined() && !t3[i].hand_back &&
                 f32c(W2(i, 0)).is_same(W2(i, 0)) && f32c(W2(i, 1)).is_same(W2(i, 1));
        }
        if (!fast) break;
        g2[i] = rows_of(grads[i], a * c);
        fast = lyc_lokr_wgrad_deferrable(cptr(g2[i]), cptr(rows), rows.size(0), (int)a, (int)b, (int)c, (int)d, code) != 0;
        if (!fast) break;
        pl[i] = F == 2 ? planes_for(W2(i, 0), x.scalar_type(), stream_of(x)) : planes_for_lr(W2(i, 0), W2(i, 1), x.scalar_type(), stream_of(x));
        fast = pl[i].defined();
      }
    }
    if (fast) {
      const int64_t M = rows.size(0);
      const int64_t nbytes = lyc_lokr_bwd_workspace_bytes(M, (int)a, (int)b, (int)c, (int)d, code);
      TORCH_CHECK(nbytes > 0, "lycoris_amd: deferrable layer without a dw1 workspace");
      std::vector<Tensor> dxs(n), wss(n), f1s(n);
      std::vector<LycLokrLinearGroupItem> items(n);
      // n <= 4 (every sibling set the modules form): the n results are summed in registers and stored once (kron4_sum_kernel)
      const bool in_kernel_sum = nx && n <= 4;
      Tensor dx_sum;
      if (in_kernel_sum) dx_sum = at::empty(rows.sizes(), x.options());
      for (size_t i = 0; i < n; ++i) {
        f1s[i] = f32c(W1(i));
        if (!in_kernel_sum) dxs[i] = at::empty(rows.sizes(), x.options());
        wss[i] = at::empty({nbytes}, x.options().dtype(at::kByte));
        items[i] = LycLokrLinearGroupItem{cptr(g2[i]), cfp(f1s[i]), planes_bwd_ptr(pl[i], c, d, 1), cptr(rows), in_kernel_sum ? nullptr : mptr(dxs[i]),
                                          mptr(wss[i]), M, (float)alphas[i]};
      }
      if (in_kernel_sum)
        check_rc(lyc_lokr_linear_bwd_group_sum(items.data(), (int)n, (int)a, (int)b, (int)c, (int)d, mptr(dx_sum), code, stream_of(x)),
                 "lyc_lokr_linear_bwd_group_sum");
      else
        check_rc(lyc_lokr_linear_bwd_group(items.data(), (int)n, (int)a, (int)b, (int)c, (int)d, code, stream_of(x)), "lyc_lokr_linear_bwd_group");
      for (size_t i = 0; i < n; ++i) {
        if (F == 2) {
          park_deferred(DeferredLokr{g2[i], rows, f1s[i], W1(i), W2(i, 0), t1[i].buf, t2[i].buf, wss[i], M, (int)a, (int)b, (int)c,
                                     (int)d, code, (float)alphas[i], stream_of(x), x.device().index()});
        } else {
          DeferredLokr item{g2[i], rows, f1s[i], W1(i), Tensor(), t1[i].buf, Tensor(), wss[i], M, (int)a, (int)b, (int)c, (int)d, code,
                            (float)alphas[i], stream_of(x), x.device().index()};
          item.w2a = W2(i, 0); item.w2b = W2(i, 1); item.d_w2a = t2[i].buf; item.d_w2b = t3[i].buf;
          park_deferred(std::move(item));
        }
      }
      if (in_kernel_sum) {
        out[0] = shaped_like(dx_sum, x);
      } else if (nx) {  // more than 4 problems: the n results summed in ONE pass per 4 sources, fp32 accumulation (lyc_sum_rows)
        for (size_t lo = 1; lo < n; lo += 3) {
          const void* src[4] = {cptr(dxs[0]), nullptr, nullptr, nullptr};
          int cnt = 1;
          for (size_t i = lo; i < n && cnt < 4; ++i) src[cnt++] = cptr(dxs[i]);
          check_rc(lyc_sum_rows(src, cnt, mptr(dxs[0]), dxs[0].numel(), code, stream_of(x)), "lyc_sum_rows");
        }
        out[0] = shaped_like(dxs[0], x);
      }
    } else {  // problem by problem, through the single-layer backward (deferred where it can be)
      Tensor dx_sum;
      for (size_t i = 0; i < n; ++i) {
        if (!grads[i].defined()) continue;
        Tensor dx;
        if (F == 3) {
          TORCH_CHECK(eager_cuda(grads[i]) && eager_cuda(x), "lycoris_amd::lokr_linear_lr_group is an eager op");
          auto [rx, r1, ra, rb] = lokr_linear_lr_bwd_one(grads[i], x, W1(i), W2(i, 0), W2(i, 1), alphas[i], nx, ctx->needs_input_grad(1 + 3 * i),
                                                         ctx->needs_input_grad(2 + 3 * i), ctx->needs_input_grad(3 + 3 * i));
          dx = rx;
          out[1 + 3 * i] = r1;
          out[2 + 3 * i] = ra;
          out[3 + 3 * i] = rb;
        } else {
          const Tensor &w1 = W1(i), &w2 = W2(i, 0);
          const bool n1 = ctx->needs_input_grad(1 + 2 * i), n2 = ctx->needs_input_grad(2 + 2 * i);
          Tensor d1, d2;
          if (eager_cuda(grads[i]) && eager_cuda(x)) {
            GradTarget u1 = grad_target(w1, n1), u2 = grad_target(w2, n2);
            bool done = false;
            if (g_defer.enabled && u2.buf.defined() && !u2.hand_back && !(u1.buf.defined() && u1.hand_back))
              done = lokr_linear_bwd_deferred(grads[i], x, w1, w2, alphas[i], nx, u1.buf, u2.buf, dx);
            if (!done) {
              dx = lokr_linear_bwd_into(grads[i], x, w1, w2, alphas[i], nx, u1.buf, u2.buf);
              d1 = finish_grad(w1, u1);
              d2 = finish_grad(w2, u2);
            }
          } else {
            static auto op = c10::Dispatcher::singleton().findSchemaOrThrow("lycoris_amd::_lokr_linear_backward", "")
                                 .typed<std::tuple<Tensor, Tensor, Tensor>(const Tensor&, const Tensor&, const Tensor&, const Tensor&,
                                                                           double, bool, bool, bool)>();
            auto [rx, r1, r2] = op.call(grads[i], x, w1, w2, alphas[i], nx, n1, n2);
            if (nx) dx = rx;
            if (n1) d1 = r1;
            if (n2) d2 = r2;
          }
          out[1 + 2 * i] = d1;
          out[2 + 2 * i] = d2;
        }
        if (dx.defined()) dx_sum = dx_sum.defined() ? dx_sum + dx : dx;
      }
      if (nx) out[0] = dx_sum;
    }
    if (has_base)
      for (size_t i = 0; i < n; ++i)
        if (ctx->needs_input_grad(1 + F * n + i)) out[1 + F * n + i] = grads[i];  // d(base + delta)/d base = 1
    if (own && nx) {  // dx += g_i W_i: the frozen layers' input gradients, accumulated by the library GEMM's own epilogue
      const int64_t I = x.size(-1);
      Tensor acc;
      if (out[0].defined()) acc = out[0].is_contiguous() ? out[0].view({-1, I}) : Tensor();
      if (out[0].defined() && !acc.defined()) {
        out[0] = out[0].contiguous();
        acc = out[0].view({-1, I});
      }
      for (size_t i = 0; i < n; ++i) {
        if (!grads[i].defined()) continue;
        const Tensor W = ctx->saved_data["lyc_frozen_w"].toTensorList().get(i);
        Tensor gi = rows_of(grads[i], W.size(0));
        if (acc.defined()) {
          acc.addmm_(gi, W);
        } else {
          acc = at::mm(gi, W);
          out[0] = shaped_like(acc, x);
        }
      }
    }
    out.resize(out.size() + 4);  // `alphas` and `F` are non-tensor inputs (surplus undefined entries are dropped by the engine)
    return out;
  }
};

std::vector<Tensor> lokr_group_autograd_impl(const Tensor& x, at::TensorList factors, at::ArrayRef<double> alphas, at::TensorList bases, int64_t F) {
  const GradAtApply ga_;
  variable_list vars;
  vars.reserve(1 + factors.size() + bases.size());
  vars.push_back(amp(x));
  for (const Tensor& t : factors) vars.push_back(t);
  for (const Tensor& t : bases) vars.push_back(t);
  return LokrLinearGroupFn::apply(at::TensorList(vars), alphas.vec(), F);  // (a TensorList: a std::vector<Tensor> argument is not seen as variables)
}
std::vector<Tensor> lokr_linear_group_autograd(const Tensor& x, at::TensorList factors, at::ArrayRef<double> alphas, at::TensorList bases) {
  return lokr_group_autograd_impl(x, factors, alphas, bases, 2);
}
std::vector<Tensor> lokr_linear_lr_group_autograd(const Tensor& x, at::TensorList factors, at::ArrayRef<double> alphas, at::TensorList bases) {
  return lokr_group_autograd_impl(x, factors, alphas, bases, 3);
}
// round 6: the frozen nn.Linear layers of the set inside the node (n >= 1; a single layer is a set of one).  factors = [w1_0, w2_0, ...]
std::vector<Tensor> lokr_adapted_linear_autograd(const Tensor& x, at::TensorList factors, at::ArrayRef<double> alphas, at::TensorList weights,
                                                 const c10::List<c10::optional<Tensor>>& biases) {
  const GradAtApply ga_;
  const size_t n = alphas.size();
  TORCH_CHECK(weights.size() == n && biases.size() == n && factors.size() == 2 * n, "lokr_adapted_linear: n alphas, 2n factors, n weights, n biases");
  variable_list vars;
  vars.reserve(1 + 2 * n);
  vars.push_back(amp(x));
  for (const Tensor& t : factors) vars.push_back(t);
  std::vector<Tensor> frozen(weights.begin(), weights.end());
  for (size_t i = 0; i < n; ++i) {
    c10::optional<Tensor> b = biases.get(i);
    frozen.push_back(b.has_value() ? *b : Tensor());
  }
  return LokrLinearGroupFn::apply(at::TensorList(vars), alphas.vec(), (int64_t)2, std::move(frozen));
}
std::vector<Tensor> lokr_linear_group_meta(const Tensor& x, at::TensorList factors, at::ArrayRef<double> alphas, at::TensorList bases) {
  auto oshape = x.sym_sizes().vec();
  oshape.back() = factors[0].sym_size(0) * factors[1].sym_size(0);
  std::vector<Tensor> ys;
  for (size_t i = 0; i < alphas.size(); ++i) ys.push_back(x.new_empty_symint(oshape));
  return ys;
}

// =====================================================================================================================
// LoKr on nn.Linear with a low-rank w2 = w2a @ w2b (reference modules/lokr.py:131-136, 370; functional/lokr.py:124-151)
// =====================================================================================================================
// The product is never formed as a tensor on the fast path: the operand planes are packed straight from the two factors
// (once per optimizer step, planes_for_lr), and the weight gradient dW2 goes to a scratch [c, d] and through the product's chain
// rule with one grouped kernel -- for parked layers at the end of the backward pass, together with their dW2.
Tensor lokr_linear_lr_fwd(const Tensor& x, const Tensor& w1, const Tensor& w2a, const Tensor& w2b, double alpha,
                          const c10::optional<Tensor>& base) {
  require_device(x, "input");
  TORCH_CHECK(w1.dim() == 2 && w2a.dim() == 2 && w2b.dim() == 2 && w2a.size(1) == w2b.size(0), "lokr_linear_lr: w1 [a, b], w2a [c, r], w2b [r, d]");
  const c10::DeviceGuard guard(x.device());
  const int64_t a = w1.size(0), b = w1.size(1), c = w2a.size(0), d = w2b.size(1);
  TORCH_CHECK(x.size(-1) == b * d, "adapter expects ", b * d, " input features, got ", x.sizes());
  const int code = dtype_code(x.scalar_type());
  Tensor rows = rows_of(x, b * d);
  Tensor pl;
  if (lyc_lokr_linear_planes_ok(rows.size(0), (int)a, (int)b, (int)c, (int)d, code) && (reinterpret_cast<uintptr_t>(cptr(rows)) & 15u) == 0)
    pl = planes_for_lr(w2a, w2b, x.scalar_type(), stream_of(x), 1, /*fwd_role=*/true);
  if (!pl.defined()) return lokr_linear_fwd(x, w1, at::mm(f32c(w2a), f32c(w2b)), alpha, base);  // shapes off the fast path
  Tensor f1 = f32c(w1);
  auto oshape = x.sizes().vec();
  oshape.back() = a * c;
  Tensor y = at::empty({rows.size(0), a * c}, x.options());
  Tensor bs;
  if (base.has_value() && base->defined()) {
    TORCH_CHECK(base->scalar_type() == x.scalar_type() && base->numel() == y.numel() && base->is_contiguous(),
                "lokr_linear_lr: `base` must be the frozen layer's contiguous output in the activation dtype");
    bs = *base;
  }
  check_rc(lyc_lokr_linear_fwd_planes(cptr(rows), cfp(f1), cptr(pl), cptr(bs), mptr(y), rows.size(0), (int)a, (int)b, (int)c, (int)d,
                                      (float)alpha, code, stream_of(x)), "lyc_lokr_linear_fwd_planes");
  return y.view(oshape);
}

// the backward of ONE low-rank layer (dx now; the factor gradients parked for the grouped launches where they can be, else dW2 into a
// scratch + the chain rule): shared by LokrLinearLrFn and the problem-by-problem path of the sibling-group node
std::tuple<Tensor, Tensor, Tensor, Tensor> lokr_linear_lr_bwd_one(const Tensor& g, const Tensor& x, const Tensor& w1, const Tensor& w2a,
                                                                  const Tensor& w2b, double alpha, bool nx, bool n1, bool na, bool nb2) {
  const c10::DeviceGuard guard(x.device());
  const int64_t a = w1.size(0), b = w1.size(1), c = w2a.size(0), d = w2b.size(1), r = w2a.size(1);
  const int code = dtype_code(x.scalar_type());
  Tensor rows = rows_of(x, b * d), g2 = rows_of(g, a * c), f1 = f32c(w1), fa = f32c(w2a), fb = f32c(w2b);
  GradTarget t1 = grad_target(w1, n1);
  const bool want_w2 = na || nb2;
  GradTarget ta = grad_target(w2a, want_w2), tb = grad_target(w2b, want_w2);
  const bool fast = lyc_lokr_linear_planes_ok(rows.size(0), (int)a, (int)b, (int)c, (int)d, code) &&
                    (reinterpret_cast<uintptr_t>(cptr(g2)) & 15u) == 0;
  Tensor pl = fast ? planes_for_lr(w2a, w2b, x.scalar_type(), stream_of(x)) : Tensor();
  const bool want_dx = nx || t1.buf.defined();
  Tensor dx, ws;
  if (want_dx) dx = at::empty(rows.sizes(), x.options());
  if (t1.buf.defined()) {
    const int64_t nbytes = lyc_lokr_bwd_workspace_bytes(rows.size(0), (int)a, (int)b, (int)c, (int)d, code);
    if (nbytes > 0) ws = at::empty({nbytes}, x.options().dtype(at::kByte));
  }
  // training configuration: every gradient goes straight into .grad -> dx now, dW1 / dW2 / chain in the grouped launches
  const bool defer = g_defer.enabled && pl.defined() && want_w2 && !ta.hand_back && !tb.hand_back && !(t1.buf.defined() && t1.hand_back) &&
                     fa.is_same(w2a) && fb.is_same(w2b) &&
                     lyc_lokr_wgrad_deferrable(cptr(g2), cptr(rows), rows.size(0), (int)a, (int)b, (int)c, (int)d, code) &&
                     (!t1.buf.defined() || ws.defined());
  if (defer) {
    if (want_dx)
      check_rc(lyc_lokr_linear_bwd_planes(cptr(g2), cptr(rows), cfp(f1), planes_bwd_ptr(pl, c, d, 1), mptr(dx), mfp(t1.buf), nullptr,
                                          mptr(ws), rows.size(0), (int)a, (int)b, (int)c, (int)d, (float)alpha, code | LYC_DEFER_WGRAD,
                                          stream_of(x)), "lyc_lokr_linear_bwd_planes(dx)");
    DeferredLokr item{g2, rows, f1, w1, Tensor(), t1.buf, Tensor(), ws, rows.size(0), (int)a, (int)b, (int)c, (int)d, code, (float)alpha,
                      stream_of(x), x.device().index()};
    item.w2a = w2a; item.w2b = w2b; item.d_w2a = ta.buf; item.d_w2b = tb.buf;
    park_deferred(std::move(item));
    return {nx ? shaped_like(dx, x) : Tensor(), Tensor(), Tensor(), Tensor()};
  }
  // immediate: dW2 into a scratch, then the chain rule (one item)
  Tensor dw2 = want_w2 ? at::zeros({c, d}, x.options().dtype(at::kFloat)) : Tensor();
  if (pl.defined()) {
    check_rc(lyc_lokr_linear_bwd_planes(cptr(g2), cptr(rows), cfp(f1), planes_bwd_ptr(pl, c, d, 1), mptr(dx), mfp(t1.buf), mfp(dw2),
                                        mptr(ws), rows.size(0), (int)a, (int)b, (int)c, (int)d, (float)alpha, code, stream_of(x)),
             "lyc_lokr_linear_bwd_planes");
  } else {
    Tensor f2 = at::mm(fa, fb);
    check_rc(lyc_lokr_linear_bwd(cptr(g2), cptr(rows), cfp(f1), cfp(f2), mptr(dx), mfp(t1.buf), mfp(dw2), mptr(ws), rows.size(0), (int)a,
                                 (int)b, (int)c, (int)d, (float)alpha, code, stream_of(x)), "lyc_lokr_linear_bwd");
  }
  if (want_w2) {
    LycLokrLrChainItem ci{cfp(dw2), cfp(fa), cfp(fb), mfp(ta.buf), mfp(tb.buf), (int)c, (int)d, (int)r};
    check_rc(lyc_lokr_lr_chain_group(&ci, 1, stream_of(x)), "lyc_lokr_lr_chain_group");
  }
  return {nx ? shaped_like(dx, x) : Tensor(), finish_grad(w1, t1), finish_grad(w2a, ta), finish_grad(w2b, tb)};
}

struct LokrLinearLrFn : public torch::autograd::Function<LokrLinearLrFn> {
  static Tensor forward(AutogradContext* ctx, const Tensor& x, const Tensor& w1, const Tensor& w2a, const Tensor& w2b, double alpha,
                        const c10::optional<Tensor>& base) {
    at::AutoDispatchBelowADInplaceOrView guard;
    TORCH_CHECK(eager_cuda(x), "lycoris_amd::lokr_linear_lr is an eager op (trace the product w2a @ w2b through lokr_linear instead)");
    Tensor y = lokr_linear_lr_fwd(x, w1, w2a, w2b, alpha, base);
    expect(w1, x);
    expect(w2a, x);
    expect(w2b, x);
    save_vars(ctx, {x, w1, w2a, w2b});
    ctx->saved_data["alpha"] = alpha;
    ctx->saved_data["has_base"] = base.has_value() && base->defined();
    return y;
  }
  static variable_list backward(AutogradContext* ctx, variable_list grads) {
    planes_mark_dirty_after_backward();  // the end of this backward pass is a step boundary for the plane cache (every LoKr node, on every path)
    auto saved = saved_vars(ctx);
    const Tensor &x = saved[0], &w1 = saved[1], &w2a = saved[2], &w2b = saved[3];
    const double alpha = ctx->saved_data["alpha"].toDouble();
    const bool nx = ctx->needs_input_grad(0), n1 = ctx->needs_input_grad(1), na = ctx->needs_input_grad(2), nb2 = ctx->needs_input_grad(3);
    const bool nbase = ctx->saved_data["has_base"].toBool() && ctx->needs_input_grad(4);
    auto [dx, d1, da, db] = lokr_linear_lr_bwd_one(grads[0], x, w1, w2a, w2b, alpha, nx, n1, na, nb2);
    return {dx, d1, da, db, Tensor(), nbase ? grads[0] : Tensor()};
  }
};
// FakeTensor / meta / functionalised inputs (torch.compile tracing): the product is formed with differentiable ATen ops and the
// traceable full-matrix op runs -- same function, no eager-only state (plane cache, park lists) touched
Tensor lokr_linear_composite(const Tensor& x, const Tensor& w1, const Tensor& w2, double alpha, const c10::optional<Tensor>& base) {
  static auto op = c10::Dispatcher::singleton().findSchemaOrThrow("lycoris_amd::lokr_linear", "")
                       .typed<Tensor(const Tensor&, const Tensor&, const Tensor&, double, const c10::optional<Tensor>&)>();
  return op.call(x, w1, w2, alpha, base);
}
Tensor lokr_linear_lr_autograd(const Tensor& x, const Tensor& w1, const Tensor& w2a, const Tensor& w2b, double alpha,
                               const c10::optional<Tensor>& base) {
  if (!eager_cuda(x)) return lokr_linear_composite(x, w1, at::matmul(w2a, w2b), alpha, base);
  const GradAtApply ga_;
  return LokrLinearLrFn::apply(amp(x), w1, w2a, w2b, alpha, base);
}
Tensor lokr_linear_lr_meta(const Tensor& x, const Tensor& w1, const Tensor& w2a, const Tensor& w2b, double alpha,
                           const c10::optional<Tensor>& base) {
  auto oshape = x.sym_sizes().vec();
  oshape.back() = w1.sym_size(0) * w2a.sym_size(0);
  return x.new_empty_symint(oshape);
}

// ---- decompose_both (reference modules/lokr.py:94-104, 358-381): w1 = w1a [a, r] @ w1b [r, b] as well ---------------------------------
// The 8 x 8 product is formed once per call below autograd (one tiny launch); its gradient dW1 lands in a scratch and goes through
// the same grouped chain-rule kernel as dW2 (c := a, d := b), so all four factors stay leaves: fused accumulation, deferral and the
// DP callback apply.  Layers off the plane kernels' fast path keep the autograd-visible products (lycoris_amd/ops.py).
struct LokrLinearLr2Fn : public torch::autograd::Function<LokrLinearLr2Fn> {
  static Tensor forward(AutogradContext* ctx, const Tensor& x, const Tensor& w1a, const Tensor& w1b, const Tensor& w2a, const Tensor& w2b,
                        double alpha, const c10::optional<Tensor>& base) {
    at::AutoDispatchBelowADInplaceOrView guard;
    TORCH_CHECK(eager_cuda(x), "lycoris_amd::lokr_linear_lr2 is an eager op");
    TORCH_CHECK(w1a.dim() == 2 && w1b.dim() == 2 && w1a.size(1) == w1b.size(0), "lokr_linear_lr2: w1a [a, r], w1b [r, b]");
    const c10::DeviceGuard dg(x.device());
    Tensor w1 = at::mm(f32c(w1a), f32c(w1b));  // grad mode is off inside Function::forward: a plain tensor
    Tensor y = lokr_linear_lr_fwd(x, w1, w2a, w2b, alpha, base);
    for (const Tensor* f : {&w1a, &w1b, &w2a, &w2b}) expect(*f, x);
    save_vars(ctx, {x, w1a, w1b, w2a, w2b});
    ctx->saved_data["alpha"] = alpha;
    ctx->saved_data["has_base"] = base.has_value() && base->defined();
    ctx->saved_data["w1"] = w1;
    return y;
  }
  static variable_list backward(AutogradContext* ctx, variable_list grads) {
    planes_mark_dirty_after_backward();  // the end of this backward pass is a step boundary for the plane cache (every LoKr node, on every path)
    auto saved = saved_vars(ctx);
    const Tensor &x = saved[0], &w1a = saved[1], &w1b = saved[2], &w2a = saved[3], &w2b = saved[4];
    const Tensor w1 = ctx->saved_data["w1"].toTensor();
    const double alpha = ctx->saved_data["alpha"].toDouble();
    const bool nx = ctx->needs_input_grad(0);
    const bool nbase = ctx->saved_data["has_base"].toBool() && ctx->needs_input_grad(5);
    const Tensor& g = grads[0];
    const c10::DeviceGuard guard(x.device());
    const int64_t a = w1.size(0), b = w1.size(1), c = w2a.size(0), d = w2b.size(1), r = w2a.size(1), r1 = w1a.size(1);
    const int code = dtype_code(x.scalar_type());
    Tensor rows = rows_of(x, b * d), g2 = rows_of(g, a * c), fa = f32c(w2a), fb = f32c(w2b), f1a = f32c(w1a), f1b = f32c(w1b);
    const bool want_w1 = ctx->needs_input_grad(1) || ctx->needs_input_grad(2);
    const bool want_w2 = ctx->needs_input_grad(3) || ctx->needs_input_grad(4);
    GradTarget t1a = grad_target(w1a, want_w1), t1b = grad_target(w1b, want_w1), ta = grad_target(w2a, want_w2), tb = grad_target(w2b, want_w2);
    const bool fast = lyc_lokr_linear_planes_ok(rows.size(0), (int)a, (int)b, (int)c, (int)d, code) &&
                      (reinterpret_cast<uintptr_t>(cptr(g2)) & 15u) == 0;
    Tensor pl = fast ? planes_for_lr(w2a, w2b, x.scalar_type(), stream_of(x)) : Tensor();
    const bool want_dx = nx || want_w1;
    Tensor dx, ws;
    if (want_dx) dx = at::empty(rows.sizes(), x.options());
    if (want_w1) {
      const int64_t nbytes = lyc_lokr_bwd_workspace_bytes(rows.size(0), (int)a, (int)b, (int)c, (int)d, code);
      if (nbytes > 0) ws = at::empty({nbytes}, x.options().dtype(at::kByte));
    }
    const bool defer = g_defer.enabled && pl.defined() && want_w1 && want_w2 && !ta.hand_back && !tb.hand_back && !t1a.hand_back && !t1b.hand_back &&
                       fa.is_same(w2a) && fb.is_same(w2b) && f1a.is_same(w1a) && f1b.is_same(w1b) && ws.defined() &&
                       lyc_lokr_wgrad_deferrable(cptr(g2), cptr(rows), rows.size(0), (int)a, (int)b, (int)c, (int)d, code);
    if (defer) {
      // dx now; dW1 partials stay in ws; dW1 / dW2 and both chain rules in the grouped launches.  On the deferrable (kron3) path the
      // dw1 pointer of the dx launch is only a "w1 gradient wanted" flag (the partials go to ws, LYC_DEFER_WGRAD skips the
      // reduction): ws itself serves as the non-null address
      check_rc(lyc_lokr_linear_bwd_planes(cptr(g2), cptr(rows), cfp(w1), planes_bwd_ptr(pl, c, d, 1), mptr(dx),
                                          reinterpret_cast<float*>(mptr(ws)), nullptr, mptr(ws),
                                          rows.size(0), (int)a, (int)b, (int)c, (int)d, (float)alpha, code | LYC_DEFER_WGRAD, stream_of(x)),
               "lyc_lokr_linear_bwd_planes(dx)");
      DeferredLokr item{g2, rows, w1, Tensor(), Tensor(), t1a.buf, Tensor(), ws, rows.size(0), (int)a, (int)b, (int)c, (int)d, code, (float)alpha,
                        stream_of(x), x.device().index()};
      item.w2a = w2a; item.w2b = w2b; item.d_w2a = ta.buf; item.d_w2b = tb.buf;
      item.w1a = w1a; item.w1b = w1b; item.d_w1a = t1a.buf; item.d_w1b = t1b.buf;
      park_deferred(std::move(item));
      return {nx ? shaped_like(dx, x) : Tensor(), Tensor(), Tensor(), Tensor(), Tensor(), Tensor(), nbase ? g : Tensor()};
    }
    Tensor dw1 = want_w1 ? at::zeros({a, b}, x.options().dtype(at::kFloat)) : Tensor();
    Tensor dw2 = want_w2 ? at::zeros({c, d}, x.options().dtype(at::kFloat)) : Tensor();
    if (pl.defined()) {
      check_rc(lyc_lokr_linear_bwd_planes(cptr(g2), cptr(rows), cfp(w1), planes_bwd_ptr(pl, c, d, 1), mptr(dx), mfp(dw1), mfp(dw2), mptr(ws),
                                          rows.size(0), (int)a, (int)b, (int)c, (int)d, (float)alpha, code, stream_of(x)),
               "lyc_lokr_linear_bwd_planes");
    } else {
      Tensor f2 = at::mm(fa, fb);
      check_rc(lyc_lokr_linear_bwd(cptr(g2), cptr(rows), cfp(w1), cfp(f2), mptr(dx), mfp(dw1), mfp(dw2), mptr(ws), rows.size(0), (int)a, (int)b,
                                   (int)c, (int)d, (float)alpha, code, stream_of(x)), "lyc_lokr_linear_bwd");
    }
    std::vector<LycLokrLrChainItem> chain;
    if (want_w2) chain.push_back(LycLokrLrChainItem{cfp(dw2), cfp(fa), cfp(fb), mfp(ta.buf), mfp(tb.buf), (int)c, (int)d, (int)r, 1});
    if (want_w1) chain.push_back(LycLokrLrChainItem{cfp(dw1), cfp(f1a), cfp(f1b), mfp(t1a.buf), mfp(t1b.buf), (int)a, (int)b, (int)r1, 1});
    if (!chain.empty()) check_rc(lyc_lokr_lr_chain_group(chain.data(), (int)chain.size(), stream_of(x)), "lyc_lokr_lr_chain_group");
    return {nx ? shaped_like(dx, x) : Tensor(), finish_grad(w1a, t1a), finish_grad(w1b, t1b), finish_grad(w2a, ta), finish_grad(w2b, tb), Tensor(),
            nbase ? g : Tensor()};
  }
};
Tensor lokr_linear_lr2_autograd(const Tensor& x, const Tensor& w1a, const Tensor& w1b, const Tensor& w2a, const Tensor& w2b, double alpha,
                                const c10::optional<Tensor>& base) {
  if (!eager_cuda(x)) return lokr_linear_composite(x, at::matmul(w1a, w1b), at::matmul(w2a, w2b), alpha, base);
  const GradAtApply ga_;
  return LokrLinearLr2Fn::apply(amp(x), w1a, w1b, w2a, w2b, alpha, base);
}
Tensor lokr_linear_lr2_cuda(const Tensor& x, const Tensor& w1a, const Tensor& w1b, const Tensor& w2a, const Tensor& w2b, double alpha,
                            const c10::optional<Tensor>& base) {
  return lokr_linear_lr_fwd(x, at::mm(f32c(w1a), f32c(w1b)), w2a, w2b, alpha, base);
}
Tensor lokr_linear_lr2_meta(const Tensor& x, const Tensor& w1a, const Tensor& w1b, const Tensor& w2a, const Tensor& w2b, double alpha,
                            const c10::optional<Tensor>& base) {
  auto oshape = x.sym_sizes().vec();
  oshape.back() = w1a.sym_size(0) * w2a.sym_size(0);
  return x.new_empty_symint(oshape);
}

// =====================================================================================================================
// LoCon on nn.Linear
// =====================================================================================================================
std::tuple<Tensor, Tensor> locon_linear_fwd(const Tensor& x, const Tensor& down, const Tensor& up, double alpha) {
  require_device(x, "input");
  const c10::DeviceGuard guard(x.device());
  // (4-D factors of a 1x1 convolution, contiguous: [r, C, 1, 1] / [O, r, 1, 1] are [r, C] / [O, r] in memory; ops.locon_conv2d hands the
  // leaves over unreshaped)
  auto mat = [](const Tensor& f) { return f.dim() == 2 || (f.dim() == 4 && f.size(2) == 1 && f.size(3) == 1 && f.is_contiguous()); };
  TORCH_CHECK(mat(down) && mat(up) && down.size(0) == up.size(1), "locon_linear: down [r, I], up [O, r] (or the contiguous 1x1 conv factors)");
  const int64_t r = down.size(0), I = down.size(1), O = up.size(0);
  TORCH_CHECK(x.size(-1) == I, "adapter expects ", I, " input features, got ", x.sizes());
  Tensor rows = rows_of(x, I), fd = f32c(down), fu = f32c(up);
  const int64_t M = rows.size(0);
  Tensor t = at::empty({M, r}, x.options().dtype(at::kFloat));
  Tensor y = at::empty({M, O}, x.options());
  check_rc(lyc_locon_linear_fwd(cptr(rows), cfp(fd), cfp(fu), mfp(t), mptr(y), M, (int)I, (int)O, (int)r, (float)alpha,
                                lc(dtype_code(x.scalar_type())), stream_of(x)), "lyc_locon_linear_fwd");
  auto oshape = x.sizes().vec();
  oshape.back() = O;
  return {y.view(oshape), t};
}

Tensor locon_linear_bwd_into(const Tensor& g, const Tensor& x, const Tensor& down, const Tensor& up, const Tensor& t, double alpha,
                             bool need_dx, const Tensor& dd, const Tensor& du, bool f32_rows = false) {
  const c10::DeviceGuard guard(x.device());
  const int64_t r = down.size(0), I = down.size(1), O = up.size(0);
  Tensor rows = rows_of(x, I), g2 = rows_of(g, O), fd = f32c(down), fu = f32c(up);
  const int64_t M = rows.size(0);
  Tensor dt = at::empty({M, r}, x.options().dtype(at::kFloat));
  Tensor dx = need_dx ? at::empty(rows.sizes(), f32_rows ? x.options().dtype(at::kFloat) : x.options()) : Tensor();
  check_rc(lyc_locon_linear_bwd(cptr(g2), cptr(rows), cfp(fd), cfp(fu), cfp(t), mfp(dt), mptr(dx), mfp(dd), mfp(du), M, (int)I,
                                (int)O, (int)r, (float)alpha, lc(dtype_code(x.scalar_type())) | (f32_rows ? LYC_F32_ROWS : 0), stream_of(x)),
           "lyc_locon_linear_bwd");
  return need_dx ? shaped_like(dx, x) : Tensor();
}

// dx now (the launch also writes dt), d_down / d_up later (park_deferred); false = not on the grouped fast path
bool locon_linear_bwd_deferred(const Tensor& g, const Tensor& x, const Tensor& down, const Tensor& up, const Tensor& t, double alpha,
                               bool need_dx, const Tensor& dd, const Tensor& du, Tensor& dx_out) {
  const c10::DeviceGuard guard(x.device());
  const int64_t r = down.size(0), I = down.size(1), O = up.size(0);
  Tensor rows = rows_of(x, I), g2 = rows_of(g, O);
  const int64_t M = rows.size(0);
  const int code = dtype_code(x.scalar_type());
  if (!lyc_locon_wgrad_deferrable(cptr(g2), cptr(rows), M, (int)I, (int)O, (int)r, code)) return false;
  Tensor fd = f32c(down), fu = f32c(up);
  Tensor dt = at::empty({M, r}, x.options().dtype(at::kFloat));
  Tensor dx = need_dx ? at::empty(rows.sizes(), x.options()) : Tensor();
  check_rc(lyc_locon_linear_bwd(cptr(g2), cptr(rows), cfp(fd), cfp(fu), cfp(t), mfp(dt), mptr(dx), nullptr, nullptr, M, (int)I,
                                (int)O, (int)r, (float)alpha, lc(code), stream_of(x)), "lyc_locon_linear_bwd(dx)");
  park_deferred(DeferredLocon{g2, rows, t, dt, down, up, dd, du, M, (int)I, (int)O, (int)r, code, (float)alpha, stream_of(x),
                              x.device().index()});
  dx_out = need_dx ? shaped_like(dx, x) : Tensor();
  return true;
}

std::tuple<Tensor, Tensor, Tensor> locon_linear_bwd(const Tensor& g, const Tensor& x, const Tensor& down, const Tensor& up,
                                                    const Tensor& t, double alpha, bool need_dx, bool need_dd, bool need_du) {
  Tensor dd = need_dd ? at::zeros(down.sizes(), down.options().dtype(at::kFloat)) : Tensor();
  Tensor du = need_du ? at::zeros(up.sizes(), up.options().dtype(at::kFloat)) : Tensor();
  Tensor dx = locon_linear_bwd_into(g, x, down, up, t, alpha, need_dx, dd, du);
  return {dx.defined() ? dx : at::empty({0}, x.options()), need_dd ? dd.to(down.scalar_type()) : at::empty({0}, down.options()),
          need_du ? du.to(up.scalar_type()) : at::empty({0}, up.options())};
}

struct LoconLinearFn : public torch::autograd::Function<LoconLinearFn> {
  static Tensor forward(AutogradContext* ctx, const Tensor& x, const Tensor& down, const Tensor& up, double alpha) {
    at::AutoDispatchBelowADInplaceOrView guard;
    static auto op = c10::Dispatcher::singleton().findSchemaOrThrow("lycoris_amd::_locon_linear_forward", "")
                         .typed<std::tuple<Tensor, Tensor>(const Tensor&, const Tensor&, const Tensor&, double)>();
    auto [y, t] = eager_cuda(x) ? locon_linear_fwd(x, down, up, alpha) : op.call(x, down, up, alpha);
    expect(down, x);
    expect(up, x);
    save_vars(ctx, {x, down, up, t});
    ctx->saved_data["alpha"] = alpha;
    return y;
  }
  static variable_list backward(AutogradContext* ctx, variable_list grads) {
    auto saved = saved_vars(ctx);
    const Tensor &x = saved[0], &down = saved[1], &up = saved[2], &t = saved[3];
    const double alpha = ctx->saved_data["alpha"].toDouble();
    const bool nx = ctx->needs_input_grad(0), nd = ctx->needs_input_grad(1), nu = ctx->needs_input_grad(2);
    Tensor g = grads[0];
    if (eager_cuda(g) && eager_cuda(x)) {
      GradTarget td = grad_target(down, nd), tu = grad_target(up, nu);
      const bool any = td.buf.defined() || tu.buf.defined(), handed = (td.buf.defined() && td.hand_back) || (tu.buf.defined() && tu.hand_back);
      if (g_defer.enabled && any && !handed) {  // both factor gradients go straight into .grad: dx now, the rest grouped
        Tensor dx;
        if (locon_linear_bwd_deferred(g, x, down, up, t, alpha, nx, td.buf, tu.buf, dx)) return {dx, Tensor(), Tensor(), Tensor()};
      }
      Tensor dx = locon_linear_bwd_into(g, x, down, up, t, alpha, nx, td.buf, tu.buf);
      return {dx, finish_grad(down, td), finish_grad(up, tu), Tensor()};
    }
    static auto op = c10::Dispatcher::singleton().findSchemaOrThrow("lycoris_amd::_locon_linear_backward", "")
                         .typed<std::tuple<Tensor, Tensor, Tensor>(const Tensor&, const Tensor&, const Tensor&, const Tensor&,
                                                                   const Tensor&, double, bool, bool, bool)>();
    auto [dx, dd, du] = op.call(g, x, down, up, t, alpha, nx, nd, nu);
    return {nx ? dx : Tensor(), nd ? dd : Tensor(), nu ? du : Tensor(), Tensor()};
  }
};
Tensor locon_linear_autograd(const Tensor& x, const Tensor& down, const Tensor& up, double alpha) {
  const GradAtApply ga_;
  return LoconLinearFn::apply(amp(x), down, up, alpha);
}

// ---- sibling LoCon projections of ONE input in one launch (round 5; VERDICT r4 #4b) -----------------------------------------------------
// The LoCon form of lokr_linear_group: factors = [down_0, up_0, down_1, up_1, ...] (equal shapes), one lyc_locon_linear_fwd_group launch
// (n <= 4 per launch, longer lists in chunks), ONE autograd node with n outputs; backward: one lyc_locon_linear_bwd_group launch (dx_i and
// dt_i of every problem), the n dx results summed in one pass (lyc_sum_rows), d_down / d_up parked for lyc_locon_wgrad_group.  Off the
// fused 16-bit rank-r path, or with gradients that are handed back to autograd, the problems run one by one through the single-layer
// functions.  Reference call sites: one LoConModule.forward per projection, modules/locon.py:309-332.
std::vector<Tensor> locon_linear_group_fwd_ts(const Tensor& x, at::TensorList factors, at::ArrayRef<double> alphas, std::vector<Tensor>& ts) {
  require_device(x, "input");
  const c10::DeviceGuard guard(x.device());
  const size_t n = alphas.size();
  TORCH_CHECK(n >= 1 && factors.size() == 2 * n, "locon_linear_group: n alphas, 2n factors");
  const Tensor &down0 = factors[0], &up0 = factors[1];
  TORCH_CHECK(down0.dim() == 2 && up0.dim() == 2 && down0.size(0) == up0.size(1), "locon_linear_group: down [r, I], up [O, r]");
  const int64_t r = down0.size(0), I = down0.size(1), O = up0.size(0);
  for (size_t i = 0; i < n; ++i)
    TORCH_CHECK(factors[2 * i].sizes() == down0.sizes() && factors[2 * i + 1].sizes() == up0.sizes(),
                "locon_linear_group: the problems of a group share their factor shapes");
  TORCH_CHECK(x.size(-1) == I, "adapter expects ", I, " input features, got ", x.sizes());
  Tensor rows = rows_of(x, I);
  const int64_t M = rows.size(0);
  const int code = dtype_code(x.scalar_type());
  auto oshape = x.sizes().vec();
  oshape.back() = O;
  std::vector<Tensor> ys(n), fds(n), fus(n);
  ts.assign(n, Tensor());
  bool grouped = n >= 2 && x.scalar_type() != at::kFloat;
  if (grouped) {
    std::vector<LycLoconLinearGroupItem> items(n);
    for (size_t i = 0; i < n; ++i) {
      fds[i] = f32c(factors[2 * i]);
      fus[i] = f32c(factors[2 * i + 1]);
      ts[i] = at::empty({M, r}, x.options().dtype(at::kFloat));
      ys[i] = at::empty({M, O}, x.options());
      items[i] = LycLoconLinearGroupItem{cptr(rows), cfp(fds[i]), cfp(fus[i]), mfp(ts[i]), mptr(ys[i]), M, (float)alphas[i]};
    }
    for (size_t lo = 0; lo < n && grouped; lo += 4) {
      const int cnt = (int)std::min<size_t>(4, n - lo);
      const int rc = lyc_locon_linear_fwd_group(items.data() + lo, cnt, (int)I, (int)O, (int)r, lc(code), stream_of(x));
      if (rc == LYC_ERR_UNSUPPORTED && lo == 0) grouped = false;  // nothing was launched: layer by layer below
      else check_rc(rc, "lyc_locon_linear_fwd_group");
    }
    if (grouped) {
      for (size_t i = 0; i < n; ++i) ys[i] = ys[i].view(oshape);
      return ys;
    }
  }
  for (size_t i = 0; i < n; ++i) {
    auto [y, t] = locon_linear_fwd(x, factors[2 * i], factors[2 * i + 1], alphas[i]);
    ys[i] = y;
    ts[i] = t;
  }
  return ys;
}
std::vector<Tensor> locon_linear_group_fwd(const Tensor& x, at::TensorList factors, at::ArrayRef<double> alphas) {
  std::vector<Tensor> ts;
  return locon_linear_group_fwd_ts(x, factors, alphas, ts);
}

struct LoconLinearGroupFn : public torch::autograd::Function<LoconLinearGroupFn> {
  // vars = [x, down_0, up_0, ..., down_{n-1}, up_{n-1}]
  static variable_list forward(AutogradContext* ctx, at::TensorList vars, std::vector<double> alphas) {
    at::AutoDispatchBelowADInplaceOrView guard;
    const size_t n = alphas.size();
    TORCH_CHECK(vars.size() == 1 + 2 * n, "locon_linear_group: bad argument list");
    const Tensor& x = vars[0];
    TORCH_CHECK(eager_cuda(x), "lycoris_amd::locon_linear_group is an eager op (trace the projections through locon_linear instead)");
    std::vector<Tensor> ts;
    std::vector<Tensor> ys = locon_linear_group_fwd_ts(x, vars.slice(1, 2 * n), alphas, ts);
    for (size_t i = 0; i < 2 * n; ++i) expect(vars[1 + i], x);
    ctx->saved_data["alphas"] = alphas;
    ctx->saved_data["n"] = (int64_t)n;
    variable_list keep(vars.begin(), vars.end());
    for (size_t i = 1; i < keep.size(); ++i)
      if (keep[i].defined() && keep[i].is_leaf() && keep[i].requires_grad()) ctx->saved_data["lyc_gleaf" + std::to_string(i)] = keep[i];
    for (const Tensor& t : ts) keep.push_back(t);
    ctx->save_for_backward(std::move(keep));
    return ys;
  }
  static variable_list backward(AutogradContext* ctx, variable_list grads) {
    variable_list s = ctx->get_saved_variables();
    const size_t n = (size_t)ctx->saved_data["n"].toInt();
    for (size_t i = 1; i < 1 + 2 * n; ++i) {  // the parameters themselves, not the copies saved-tensor hooks hand back (see save_vars)
      auto it = ctx->saved_data.find("lyc_gleaf" + std::to_string(i));
      if (it == ctx->saved_data.end() || !it->second.isTensor()) continue;
      const Tensor& p = it->second.toTensor();
      if (p.defined() && s[i].defined() && !s[i].is_same(p) && s[i].sizes() == p.sizes()) s[i] = p;
    }
    const std::vector<double> alphas = ctx->saved_data["alphas"].toDoubleVector();
    const Tensor& x = s[0];
    const bool nx = ctx->needs_input_grad(0);
    variable_list out(1 + 2 * n);
    const c10::DeviceGuard guard(x.device());
    const int64_t r = s[1].size(0), I = s[1].size(1), O = s[2].size(0);
    const int code = dtype_code(x.scalar_type());
    auto DOWN = [&](size_t i) -> const Tensor& { return s[1 + 2 * i]; };
    auto UP = [&](size_t i) -> const Tensor& { return s[2 + 2 * i]; };
    auto T_ = [&](size_t i) -> const Tensor& { return s[1 + 2 * n + i]; };
    // ---- grouped: every problem's factor gradients go straight into .grad and the shapes are deferrable ---------------------------
    bool fast = n >= 2 && n <= 4 && g_defer.enabled && eager_cuda(x) && x.scalar_type() != at::kFloat;
    std::vector<GradTarget> td(n), tu(n);
    std::vector<Tensor> g2(n);
    Tensor rows;
    if (fast) {
      rows = rows_of(x, I);
      // lyc_sum_rows (the sum of the n dx results) moves 16-byte pieces: nothing is launched unless it can run too
      fast = !nx || (rows.numel() % 8 == 0 && reinterpret_cast<uintptr_t>(rows.data_ptr()) % 16 == 0);
      for (size_t i = 0; i < n && fast; ++i) {
        fast = grads[i].defined() && eager_cuda(grads[i]);
        if (!fast) break;
        td[i] = grad_target(DOWN(i), ctx->needs_input_grad(1 + 2 * i));
        tu[i] = grad_target(UP(i), ctx->needs_input_grad(2 + 2 * i));
        fast = td[i].buf.defined() && !td[i].hand_back && tu[i].buf.defined() && !tu[i].hand_back;
        if (!fast) break;
        g2[i] = rows_of(grads[i], O);
        fast = lyc_locon_wgrad_deferrable(cptr(g2[i]), cptr(rows), rows.size(0), (int)I, (int)O, (int)r, code) != 0;
      }
    }
    if (fast) {
      const int64_t M = rows.size(0);
      std::vector<Tensor> dxs(n), dts(n), fds(n), fus(n);
      std::vector<LycLoconLinearGroupItem> items(n);
      for (size_t i = 0; i < n; ++i) {
        fds[i] = f32c(DOWN(i));
        fus[i] = f32c(UP(i));
        dts[i] = at::empty({M, r}, x.options().dtype(at::kFloat));
        items[i] = LycLoconLinearGroupItem{cptr(g2[i]), cfp(fds[i]), cfp(fus[i]), mfp(dts[i]), nullptr, M, (float)alphas[i]};
      }
      // round 6: the shared input's gradient as ONE expand stage over the n `mid` tiles (bneck4_sum_kernel): no dx_i, no summation pass
      bool summed = false;
      if (nx) {
        Tensor dx = at::empty(rows.sizes(), x.options());
        const int rs = lyc_locon_linear_bwd_group_sum(items.data(), (int)n, (int)I, (int)O, (int)r, mptr(dx), lc(code), stream_of(x));
        if (rs != LYC_ERR_UNSUPPORTED) {
          check_rc(rs, "lyc_locon_linear_bwd_group_sum");
          summed = true;
          out[0] = shaped_like(dx, x);
        }
      }
      int rc = LYC_OK;
      if (!summed) {
        for (size_t i = 0; i < n; ++i) {
          dxs[i] = at::empty(rows.sizes(), x.options());
          items[i].out = mptr(dxs[i]);
        }
        rc = lyc_locon_linear_bwd_group(items.data(), (int)n, (int)I, (int)O, (int)r, lc(code), stream_of(x));
      }
      if (rc == LYC_ERR_UNSUPPORTED) {
        fast = false;  // nothing was launched
      } else {
        check_rc(rc, "lyc_locon_linear_bwd_group");
        for (size_t i = 0; i < n; ++i)
          park_deferred(DeferredLocon{g2[i], rows, T_(i), dts[i], DOWN(i), UP(i), td[i].buf, tu[i].buf, M, (int)I, (int)O, (int)r, code,
                                      (float)alphas[i], stream_of(x), x.device().index()});
        if (nx && !summed) {  // the shared input's gradient: the n results in ONE pass, fp32 accumulation, one more rounding
          const void* src[4] = {nullptr, nullptr, nullptr, nullptr};
          for (size_t i = 0; i < n; ++i) src[i] = cptr(dxs[i]);
          check_rc(lyc_sum_rows(src, (int)n, mptr(dxs[0]), dxs[0].numel(), code, stream_of(x)), "lyc_sum_rows");
          out[0] = shaped_like(dxs[0], x);
        }
      }
    }
    if (!fast) {  // problem by problem, through the single-layer backward (deferred where it can be)
      Tensor dx_sum;
      for (size_t i = 0; i < n; ++i) {
        if (!grads[i].defined()) continue;
        const Tensor &down = DOWN(i), &up = UP(i);
        GradTarget d = grad_target(down, ctx->needs_input_grad(1 + 2 * i)), u = grad_target(up, ctx->needs_input_grad(2 + 2 * i));
        const bool any = d.buf.defined() || u.buf.defined(), handed = (d.buf.defined() && d.hand_back) || (u.buf.defined() && u.hand_back);
        Tensor dx;
        bool done = false;
        if (g_defer.enabled && any && !handed) done = locon_linear_bwd_deferred(grads[i], x, down, up, T_(i), alphas[i], nx, d.buf, u.buf, dx);
        if (!done) {
          dx = locon_linear_bwd_into(grads[i], x, down, up, T_(i), alphas[i], nx, d.buf, u.buf);
          out[1 + 2 * i] = finish_grad(down, d);
          out[2 + 2 * i] = finish_grad(up, u);
        }
        if (dx.defined()) dx_sum = dx_sum.defined() ? dx_sum + dx : dx;
      }
      if (nx) out[0] = dx_sum;
    }
    out.resize(out.size() + 2);  // `alphas` is a non-tensor input (surplus undefined entries are dropped by the engine)
    return out;
  }
};
std::vector<Tensor> locon_linear_group_autograd(const Tensor& x, at::TensorList factors, at::ArrayRef<double> alphas) {
  const GradAtApply ga_;
  variable_list vars;
  vars.reserve(1 + factors.size());
  vars.push_back(amp(x));
  for (const Tensor& t : factors) vars.push_back(t);
  return LoconLinearGroupFn::apply(at::TensorList(vars), alphas.vec());
}
std::vector<Tensor> locon_linear_group_meta(const Tensor& x, at::TensorList factors, at::ArrayRef<double> alphas) {
  auto oshape = x.sym_sizes().vec();
  oshape.back() = factors[1].sym_size(0);
  std::vector<Tensor> ys;
  for (size_t i = 0; i < alphas.size(); ++i) ys.push_back(x.new_empty_symint(oshape));
  return ys;
}
Tensor locon_linear_cuda(const Tensor& x, const Tensor& down, const Tensor& up, double alpha) {
  return std::get<0>(locon_linear_fwd(x, down, up, alpha));
}
std::tuple<Tensor, Tensor> locon_linear_fwd_meta(const Tensor& x, const Tensor& down, const Tensor& up, double alpha) {
  auto oshape = x.sym_sizes().vec();
  oshape.back() = up.sym_size(0);
  c10::SymInt M = x.sym_numel() / x.sym_size(-1);
  return {x.new_empty_symint(oshape), x.new_empty_symint({M, down.sym_size(0)}, x.options().dtype(at::kFloat))};
}
Tensor locon_linear_meta(const Tensor& x, const Tensor& down, const Tensor& up, double alpha) {
  return std::get<0>(locon_linear_fwd_meta(x, down, up, alpha));
}
std::tuple<Tensor, Tensor, Tensor> locon_linear_bwd_meta(const Tensor& g, const Tensor& x, const Tensor& down, const Tensor& up,
                                                         const Tensor& t, double alpha, bool nx, bool nd, bool nu) {
  return {nx ? at::empty_like(x) : x.new_empty({0}), nd ? at::empty_like(down) : down.new_empty({0}),
          nu ? at::empty_like(up) : up.new_empty({0})};
}

// =====================================================================================================================
// LoHa on nn.Linear
// =====================================================================================================================
std::tuple<Tensor, Tensor> loha_linear_fwd(const Tensor& x, const Tensor& w1a, const Tensor& w1b, const Tensor& w2a,
                                           const Tensor& w2b, double alpha) {
  require_device(x, "input");
  const c10::DeviceGuard guard(x.device());
  const int64_t O = w1a.size(0), r = w1a.size(1), I = w1b.size(1);
  TORCH_CHECK(x.size(-1) == I, "adapter expects ", I, " input features, got ", x.sizes());
  Tensor rows = rows_of(x, I), a1 = f32c(w1a), b1 = f32c(w1b), a2 = f32c(w2a), b2 = f32c(w2b);
  const int code = dtype_code(x.scalar_type());
  // leaf-parameter factors: the operand plane comes from the cache (rebuilt once per optimizer step, all layers in grouped launches)
  Tensor ws = loha_plane_for(w1a, w1b, w2a, w2b, alpha, x.scalar_type(), stream_of(x), /*fwd_role=*/true);
  const bool cached = ws.defined();
  if (!cached) ws = at::empty({lyc_loha_workspace_bytes((int)O, (int)I, code)}, x.options().dtype(at::kByte));
  Tensor y = at::empty({rows.size(0), O}, x.options());
  check_rc(lyc_loha_linear_fwd(cptr(rows), cfp(a1), cfp(b1), cfp(a2), cfp(b2), mptr(ws), mptr(y), rows.size(0), (int)I, (int)O,
                               (int)r, (float)alpha, code | (cached ? LYC_PLANE_READY : 0), stream_of(x)), "lyc_loha_linear_fwd");
  auto oshape = x.sizes().vec();
  oshape.back() = O;
  return {y.view(oshape), ws};
}

Tensor loha_linear_bwd_into(const Tensor& g, const Tensor& x, const Tensor& w1a, const Tensor& w1b, const Tensor& w2a,
                            const Tensor& w2b, const Tensor& ws, double alpha, bool need_dx, Tensor (&d)[4], bool f32_rows = false) {
  const c10::DeviceGuard guard(x.device());
  const int64_t O = w1a.size(0), r = w1a.size(1), I = w1b.size(1);
  Tensor rows = rows_of(x, I), g2 = rows_of(g, O), a1 = f32c(w1a), b1 = f32c(w1b), a2 = f32c(w2a), b2 = f32c(w2b);
  const bool any = d[0].defined() || d[1].defined() || d[2].defined() || d[3].defined();
  const Tensor* fs[4] = {&w1a, &w1b, &w2a, &w2b};
  Tensor tmp[4];
  if (any)  // the factor-gradient kernel produces the four gradients as a set
    for (int i = 0; i < 4; ++i) tmp[i] = d[i].defined() ? d[i] : at::zeros(fs[i]->sizes(), fs[i]->options().dtype(at::kFloat));
  Tensor gw = any ? at::empty({O, I}, x.options().dtype(at::kFloat)) : Tensor();
  Tensor dx = need_dx ? at::empty(rows.sizes(), f32_rows ? x.options().dtype(at::kFloat) : x.options()) : Tensor();
  check_rc(lyc_loha_linear_bwd(cptr(g2), cptr(rows), cfp(a1), cfp(b1), cfp(a2), cfp(b2), cptr(ws), mfp(gw), mptr(dx), mfp(tmp[0]),
                               mfp(tmp[1]), mfp(tmp[2]), mfp(tmp[3]), rows.size(0), (int)I, (int)O, (int)r, (float)alpha,
                               dtype_code(x.scalar_type()) | (f32_rows ? LYC_F32_ROWS : 0), stream_of(x)), "lyc_loha_linear_bwd");
  return need_dx ? shaped_like(dx, x) : Tensor();
}

// dx = g dW now, G = g^T x and HadaWeight.backward later (park_deferred); false = not on the grouped path
bool loha_linear_bwd_deferred(const Tensor& g, const Tensor& x, const Tensor& w1a, const Tensor& w1b, const Tensor& w2a,
                              const Tensor& w2b, const Tensor& ws, double alpha, bool need_dx, Tensor (&d)[4], Tensor& dx_out) {
  const c10::DeviceGuard guard(x.device());
  const int64_t O = w1a.size(0), r = w1a.size(1), I = w1b.size(1);
  Tensor rows = rows_of(x, I), g2 = rows_of(g, O);
  const int code = dtype_code(x.scalar_type());
  if (!lyc_loha_wgrad_deferrable(cptr(g2), cptr(rows), rows.size(0), (int)I, (int)O, (int)r, code)) return false;
  Tensor a1 = f32c(w1a), b1 = f32c(w1b), a2 = f32c(w2a), b2 = f32c(w2b);
  Tensor dx;
  if (need_dx) {
    dx = at::empty(rows.sizes(), x.options());
    check_rc(lyc_loha_linear_bwd(cptr(g2), cptr(rows), cfp(a1), cfp(b1), cfp(a2), cfp(b2), cptr(ws), nullptr, mptr(dx), nullptr,
                                 nullptr, nullptr, nullptr, rows.size(0), (int)I, (int)O, (int)r, (float)alpha, code, stream_of(x)),
             "lyc_loha_linear_bwd(dx)");
  }
  park_deferred(DeferredLoha{g2, rows, {a1, b1, a2, b2}, {w1a, w1b, w2a, w2b}, {d[0], d[1], d[2], d[3]}, rows.size(0), (int)I,
                             (int)O, (int)r, code, (float)alpha, stream_of(x), x.device().index()});
  dx_out = need_dx ? shaped_like(dx, x) : Tensor();
  return true;
}

std::tuple<Tensor, Tensor, Tensor, Tensor, Tensor> loha_linear_bwd(const Tensor& g, const Tensor& x, const Tensor& w1a,
                                                                   const Tensor& w1b, const Tensor& w2a, const Tensor& w2b,
                                                                   const Tensor& ws, double alpha, bool need_dx, bool need_f) {
  const Tensor* fs[4] = {&w1a, &w1b, &w2a, &w2b};
  Tensor d[4];
  if (need_f)
    for (int i = 0; i < 4; ++i) d[i] = at::zeros(fs[i]->sizes(), fs[i]->options().dtype(at::kFloat));
  Tensor dx = loha_linear_bwd_into(g, x, w1a, w1b, w2a, w2b, ws, alpha, need_dx, d);
  auto out = [&](int i) { return need_f ? d[i].to(fs[i]->scalar_type()) : at::empty({0}, fs[i]->options()); };
  return {dx.defined() ? dx : at::empty({0}, x.options()), out(0), out(1), out(2), out(3)};
}

struct LohaLinearFn : public torch::autograd::Function<LohaLinearFn> {
  static Tensor forward(AutogradContext* ctx, const Tensor& x, const Tensor& w1a, const Tensor& w1b, const Tensor& w2a,
                        const Tensor& w2b, double alpha) {
    at::AutoDispatchBelowADInplaceOrView guard;
    static auto op = c10::Dispatcher::singleton().findSchemaOrThrow("lycoris_amd::_loha_linear_forward", "")
                         .typed<std::tuple<Tensor, Tensor>(const Tensor&, const Tensor&, const Tensor&, const Tensor&, const Tensor&,
                                                           double)>();
    auto [y, ws] = eager_cuda(x) ? loha_linear_fwd(x, w1a, w1b, w2a, w2b, alpha) : op.call(x, w1a, w1b, w2a, w2b, alpha);
    for (const Tensor* f : {&w1a, &w1b, &w2a, &w2b}) expect(*f, x);
    save_vars(ctx, {x, w1a, w1b, w2a, w2b, ws});
    ctx->saved_data["alpha"] = alpha;
    return y;
  }
  static variable_list backward(AutogradContext* ctx, variable_list grads) {
    planes_mark_dirty_after_backward();  // the end of this backward pass is a step boundary for the operand-plane cache (LoHa: round 6)
    auto s = saved_vars(ctx);
    const double alpha = ctx->saved_data["alpha"].toDouble();
    const bool nx = ctx->needs_input_grad(0);
    bool nf[4], any = false;
    for (int i = 0; i < 4; ++i) any = (nf[i] = ctx->needs_input_grad(1 + i)) || any;
    Tensor g = grads[0];
    if (eager_cuda(g) && eager_cuda(s[0])) {
      GradTarget t[4];
      Tensor d[4];
      for (int i = 0; i < 4; ++i) {
        t[i] = grad_target(s[1 + i], nf[i]);
        d[i] = t[i].buf;
      }
      bool all_accum = g_defer.enabled;  // all four factor gradients go straight into .grad: dx now, the rest grouped
      for (int i = 0; i < 4; ++i) all_accum = all_accum && d[i].defined() && !t[i].hand_back;
      if (all_accum) {
        Tensor dx;
        if (loha_linear_bwd_deferred(g, s[0], s[1], s[2], s[3], s[4], s[5], alpha, nx, d, dx))
          return {dx, Tensor(), Tensor(), Tensor(), Tensor(), Tensor()};
      }
      Tensor dx = loha_linear_bwd_into(g, s[0], s[1], s[2], s[3], s[4], s[5], alpha, nx, d);
      return {dx, finish_grad(s[1], t[0]), finish_grad(s[2], t[1]), finish_grad(s[3], t[2]), finish_grad(s[4], t[3]), Tensor()};
    }
    static auto op = c10::Dispatcher::singleton().findSchemaOrThrow("lycoris_amd::_loha_linear_backward", "")
                         .typed<std::tuple<Tensor, Tensor, Tensor, Tensor, Tensor>(const Tensor&, const Tensor&, const Tensor&,
                                                                                   const Tensor&, const Tensor&, const Tensor&,
                                                                                   const Tensor&, double, bool, bool)>();
    auto [dx, d0, d1, d2, d3] = op.call(g, s[0], s[1], s[2], s[3], s[4], s[5], alpha, nx, any);
    return {nx ? dx : Tensor(), nf[0] ? d0 : Tensor(), nf[1] ? d1 : Tensor(), nf[2] ? d2 : Tensor(), nf[3] ? d3 : Tensor(), Tensor()};
  }
};
Tensor loha_linear_autograd(const Tensor& x, const Tensor& w1a, const Tensor& w1b, const Tensor& w2a, const Tensor& w2b,
                            double alpha) {
  const GradAtApply ga_;
  return LohaLinearFn::apply(amp(x), w1a, w1b, w2a, w2b, alpha);
}
Tensor loha_linear_cuda(const Tensor& x, const Tensor& w1a, const Tensor& w1b, const Tensor& w2a, const Tensor& w2b, double alpha) {
  return std::get<0>(loha_linear_fwd(x, w1a, w1b, w2a, w2b, alpha));
}
std::tuple<Tensor, Tensor> loha_linear_fwd_meta(const Tensor& x, const Tensor& w1a, const Tensor& w1b, const Tensor& w2a,
                                                const Tensor& w2b, double alpha) {
  auto oshape = x.sym_sizes().vec();
  oshape.back() = w1a.sym_size(0);
  const int64_t O = w1a.size(0), I = w1b.size(1);
  return {x.new_empty_symint(oshape),
          x.new_empty({lyc_loha_workspace_bytes((int)O, (int)I, dtype_code(x.scalar_type()))}, x.options().dtype(at::kByte))};
}
Tensor loha_linear_meta(const Tensor& x, const Tensor& w1a, const Tensor& w1b, const Tensor& w2a, const Tensor& w2b, double alpha) {
  return std::get<0>(loha_linear_fwd_meta(x, w1a, w1b, w2a, w2b, alpha));
}
std::tuple<Tensor, Tensor, Tensor, Tensor, Tensor> loha_linear_bwd_meta(const Tensor& g, const Tensor& x, const Tensor& w1a,
                                                                        const Tensor& w1b, const Tensor& w2a, const Tensor& w2b,
                                                                        const Tensor& ws, double alpha, bool nx, bool nf) {
  auto e = [&](const Tensor& t, bool n) { return n ? at::empty_like(t) : t.new_empty({0}); };
  return {e(x, nx), e(w1a, nf), e(w1b, nf), e(w2a, nf), e(w2b, nf)};
}

// =====================================================================================================================
// (IA)^3 per-channel affine:  out = a * (s0 + w[c] * mult) - bias[c] * w[c] * mult   over dimension chan_dim
// =====================================================================================================================
void chan_dims(const Tensor& t, int64_t chan_dim, int64_t& outer, int64_t& C, int64_t& inner) {
  C = t.size(chan_dim);
  outer = inner = 1;
  for (int64_t i = 0; i < chan_dim; ++i) outer *= t.size(i);
  for (int64_t i = chan_dim + 1; i < t.dim(); ++i) inner *= t.size(i);
}

bool rows_are_free(const Tensor& t);
// A channels_last [B, C, H, W] tensor scaled along C IS a contiguous [B, H, W, C] tensor scaled along its last dimension: no copy
// (round 3: `.contiguous()` turned every (IA)^3 Conv2d layer of the mixed preset into two NCHW <-> NHWC copies per pass, +3.5 ms)
bool chan_is_cl(const Tensor& t, int64_t chan_dim) {
  return t.dim() == 4 && at::maybe_wrap_dim(chan_dim, 4) == 1 && rows_are_free(t);
}

Tensor chan_scale(const Tensor& a_in, const Tensor& w, const c10::optional<Tensor>& bias, double s0, double mult, int64_t chan_dim) {
  require_device(a_in, "input");
  const c10::DeviceGuard guard(a_in.device());
  const bool cl = chan_is_cl(a_in, chan_dim);
  Tensor a = cl ? a_in.permute({0, 2, 3, 1}) : a_in.contiguous();
  if (cl) chan_dim = 3;
  chan_dim = at::maybe_wrap_dim(chan_dim, a.dim());
  int64_t outer, C, inner;
  chan_dims(a, chan_dim, outer, C, inner);
  Tensor wf = f32c(w).reshape({-1});
  TORCH_CHECK(wf.numel() == C, "(IA)^3 weight has ", wf.numel(), " entries, channel dim has ", C);
  Tensor bf = (bias.has_value() && bias->defined()) ? f32c(*bias).reshape({-1}) : Tensor();
  Tensor out = at::empty_like(a);
  check_rc(lyc_chan_scale(cptr(a), cfp(wf), cfp(bf), mptr(out), outer, C, inner, (float)s0, (float)mult,
                          dtype_code(a.scalar_type()), stream_of(a)), "lyc_chan_scale");
  return cl ? out.permute({0, 3, 1, 2}) : out;
}

// dw[c] += mult * sum g * (a - bias[c])   accumulated into `dw` (fp32, C entries)
void chan_reduce_into(const Tensor& g_in, const Tensor& a_in, const c10::optional<Tensor>& bias, double mult, int64_t chan_dim,
                      const Tensor& dw) {
  const c10::DeviceGuard guard(a_in.device());
  const bool cl = chan_is_cl(a_in, chan_dim);
  Tensor a = cl ? a_in.permute({0, 2, 3, 1}) : a_in.contiguous();
  Tensor g = cl ? g_in.permute({0, 2, 3, 1}).contiguous() : g_in.contiguous();  // a channels_last g: the view itself
  if (cl) chan_dim = 3;
  chan_dim = at::maybe_wrap_dim(chan_dim, a.dim());
  int64_t outer, C, inner;
  chan_dims(a, chan_dim, outer, C, inner);
  Tensor bf = (bias.has_value() && bias->defined()) ? f32c(*bias).reshape({-1}) : Tensor();
  check_rc(lyc_chan_reduce(cptr(g), cptr(a), cfp(bf), mfp(dw), outer, C, inner, (float)mult, dtype_code(a.scalar_type()),
                           stream_of(a)), "lyc_chan_reduce");
}
// the whole backward in one pass over g (lyc_chan_bwd): da (when wanted) returned, dw accumulated into `dw` (when defined)
Tensor chan_bwd_into(const Tensor& g_in, const Tensor& a_in, const Tensor& w, const c10::optional<Tensor>& bias, double s0, double mult,
                     int64_t chan_dim, bool need_da, const Tensor& dw) {
  const c10::DeviceGuard guard(a_in.device());
  const bool cl = chan_is_cl(a_in, chan_dim);
  Tensor a = cl ? a_in.permute({0, 2, 3, 1}) : a_in.contiguous();
  Tensor g = cl ? g_in.permute({0, 2, 3, 1}).contiguous() : g_in.contiguous();
  if (cl) chan_dim = 3;
  chan_dim = at::maybe_wrap_dim(chan_dim, a.dim());
  int64_t outer, C, inner;
  chan_dims(a, chan_dim, outer, C, inner);
  Tensor wf = f32c(w).reshape({-1});
  Tensor bf = (bias.has_value() && bias->defined()) ? f32c(*bias).reshape({-1}) : Tensor();
  Tensor da = need_da ? at::empty_like(g) : Tensor();
  check_rc(lyc_chan_bwd(cptr(g), cptr(a), cfp(wf), cfp(bf), need_da ? mptr(da) : nullptr, dw.defined() ? mfp(dw) : nullptr, outer, C, inner,
                        (float)s0, (float)mult, dtype_code(a.scalar_type()), stream_of(a)), "lyc_chan_bwd");
  if (!need_da) return Tensor();
  return cl ? da.permute({0, 3, 1, 2}) : da;
}
Tensor chan_reduce(const Tensor& g, const Tensor& a, const c10::optional<Tensor>& bias, const Tensor& w_like, double mult,
                   int64_t chan_dim) {
  Tensor dw = at::zeros(w_like.sizes(), w_like.options().dtype(at::kFloat));
  chan_reduce_into(g, a, bias, mult, chan_dim, dw);
  return dw.to(w_like.scalar_type());
}

struct ChanAffineFn : public torch::autograd::Function<ChanAffineFn> {
  static Tensor forward(AutogradContext* ctx, const Tensor& a, const Tensor& w, const c10::optional<Tensor>& bias, double s0,
                        double mult, int64_t chan_dim) {
    at::AutoDispatchBelowADInplaceOrView guard;
    static auto op = c10::Dispatcher::singleton().findSchemaOrThrow("lycoris_amd::chan_affine", "")
                         .typed<Tensor(const Tensor&, const Tensor&, const c10::optional<Tensor>&, double, double, int64_t)>();
    Tensor out = op.call(a, w, bias, s0, mult, chan_dim);
    expect(w, a);
    save_vars(ctx, {a, w, bias.has_value() ? *bias : Tensor()});
    ctx->saved_data["s0"] = s0;
    ctx->saved_data["mult"] = mult;
    ctx->saved_data["chan_dim"] = at::maybe_wrap_dim(chan_dim, a.dim());
    return out;
  }
  static variable_list backward(AutogradContext* ctx, variable_list grads) {
    auto s = saved_vars(ctx);
    const Tensor &a = s[0], &w = s[1];
    c10::optional<Tensor> bias = s[2].defined() ? c10::optional<Tensor>(s[2]) : c10::nullopt;
    const double s0 = ctx->saved_data["s0"].toDouble(), mult = ctx->saved_data["mult"].toDouble();
    const int64_t chan_dim = ctx->saved_data["chan_dim"].toInt();
    Tensor g = grads[0], da, dw;
    static auto scale = c10::Dispatcher::singleton().findSchemaOrThrow("lycoris_amd::chan_affine", "")
                            .typed<Tensor(const Tensor&, const Tensor&, const c10::optional<Tensor>&, double, double, int64_t)>();
    if (eager_cuda(g) && eager_cuda(a) && g.scalar_type() == a.scalar_type() && (ctx->needs_input_grad(0) || ctx->needs_input_grad(1))) {
      // eager device tensors: da and dw in ONE pass over g (round 6)
      GradTarget t = grad_target(w, ctx->needs_input_grad(1));
      da = chan_bwd_into(g, a, w, bias, s0, mult, chan_dim, ctx->needs_input_grad(0), t.buf);
      return {da, finish_grad(w, t), Tensor(), Tensor(), Tensor(), Tensor()};
    }
    if (ctx->needs_input_grad(0)) da = scale.call(g, w, c10::nullopt, s0, mult, chan_dim);
    if (ctx->needs_input_grad(1)) {
      if (eager_cuda(g) && eager_cuda(a)) {
        GradTarget t = grad_target(w, true);
        chan_reduce_into(g, a, bias, mult, chan_dim, t.buf);
        dw = finish_grad(w, t);
      } else {
        static auto red = c10::Dispatcher::singleton().findSchemaOrThrow("lycoris_amd::_chan_reduce", "")
                              .typed<Tensor(const Tensor&, const Tensor&, const c10::optional<Tensor>&, const Tensor&, double, int64_t)>();
        dw = red.call(g, a, bias, w, mult, chan_dim);
      }
    }
    return {da, dw, Tensor(), Tensor(), Tensor(), Tensor()};
  }
};
Tensor chan_affine_autograd(const Tensor& a, const Tensor& w, const c10::optional<Tensor>& bias, double s0, double mult,
                            int64_t chan_dim) {
  const GradAtApply ga_;
  return ChanAffineFn::apply(amp(a), w, bias, s0, mult, chan_dim);
}
Tensor chan_affine_meta(const Tensor& a, const Tensor& w, const c10::optional<Tensor>& bias, double s0, double mult, int64_t chan_dim) {
  return at::empty_like(a, a.options().memory_format(at::MemoryFormat::Contiguous));
}
Tensor chan_reduce_meta(const Tensor& g, const Tensor& a, const c10::optional<Tensor>& bias, const Tensor& w_like, double mult,
                        int64_t chan_dim) {
  return at::empty_like(w_like);
}

// =====================================================================================================================
// Conv2d without im2col (LoKr, LoCon): NHWC row matrices, see include/lycoris_amd.h
// =====================================================================================================================
struct Geom {
  int kh, kw, sh, sw, ph, pw, dh, dw;
  int64_t Ho, Wo;
};
Geom geom_of(at::IntArrayRef ksize, at::IntArrayRef s, at::IntArrayRef p, at::IntArrayRef d, int64_t H, int64_t W) {
  Geom g{(int)ksize[0], (int)ksize[1], (int)s[0], (int)s[1], (int)p[0], (int)p[1], (int)d[0], (int)d[1], 0, 0};
  g.Ho = (H + 2 * g.ph - g.dh * (g.kh - 1) - 1) / g.sh + 1;
  g.Wo = (W + 2 * g.pw - g.dw * (g.kw - 1) - 1) / g.sw + 1;
  return g;
}

// a channels_last tensor already IS the NHWC row matrix (degenerate shapes are ambiguous: treated as NCHW)
bool rows_are_free(const Tensor& t) {
  return t.is_contiguous(at::MemoryFormat::ChannelsLast) && !(t.size(1) == 1 || t.size(2) * t.size(3) == 1);
}
// [B, C, H, W] -> [B*H*W, C] NHWC rows; *copied = false when the tensor already is such a matrix (channels_last)
Tensor rows_view(const Tensor& t, bool* copied) {
  const int64_t B = t.size(0), C = t.size(1), H = t.size(2), W = t.size(3);
  if (rows_are_free(t)) {
    *copied = false;
    return t.permute({0, 2, 3, 1}).reshape({B * H * W, C});
  }
  Tensor tc = t.contiguous();
  Tensor rows = at::empty({B * H * W, C}, t.options());
  check_rc(lyc_nchw_to_rows(cptr(tc), mptr(rows), B, C, H * W, dtype_code(t.scalar_type()), stream_of(t)), "lyc_nchw_to_rows");
  *copied = true;
  return rows;
}
Tensor from_rows(const Tensor& rows, int64_t B, int64_t H, int64_t W, bool channels_last) {
  const int64_t C = rows.size(1);
  if (channels_last) return rows.view({B, H, W, C}).permute({0, 3, 1, 2});
  Tensor out = at::empty({B, C, H, W}, rows.options());
  check_rc(lyc_rows_to_nchw(cptr(rows), mptr(out), B, C, H * W, dtype_code(rows.scalar_type()), stream_of(rows)), "lyc_rows_to_nchw");
  return out;
}

// gradient buffer of a 4-D parameter whose kernels work in the permute(0, 2, 3, 1) (window-major) layout
GradTarget cl_grad_target(const Tensor& p, bool need, at::IntArrayRef shape_p) {
  GradTarget g;
  if (!need) return g;
  if (g_accum.enabled && p.is_leaf()) {
    const Tensor& gr = p.grad();
    if (gr.defined() && gr.scalar_type() == at::kFloat && gr.device() == p.device() && gr.permute({0, 2, 3, 1}).is_contiguous()) {
      g.buf = gr.permute({0, 2, 3, 1});
      return g;
    }
  }
  g.buf = at::zeros(shape_p, p.options().dtype(at::kFloat));
  g.hand_back = true;
  return g;
}
Tensor cl_finish(const Tensor& p, const GradTarget& g) {
  if (!g.buf.defined()) return Tensor();
  if (g.hand_back) return g.buf.permute({0, 3, 1, 2}).to(p.scalar_type());
  notify(p);
  return Tensor();
}

Tensor conv2d_meta(const Tensor& x, const Tensor& f0, const Tensor& f1, double alpha, at::IntArrayRef stride, at::IntArrayRef padding,
                   at::IntArrayRef dilation, bool lokr);

// ---- the Conv2d ops as functional forward / backward dispatcher ops (what torch.compile traces: FakeTensors see the Meta
// kernels, the compiled graph calls the CUDA kernels below).  The eager autograd Functions above keep the NHWC row matrix of
// x between forward and backward; these recompute it (one transpose for an NCHW tensor, nothing for channels_last).
Tensor lokr_conv2d_fwd(const Tensor& x, const Tensor& w1, const Tensor& w2, double alpha, at::IntArrayRef stride,
                       at::IntArrayRef padding, at::IntArrayRef dilation) {
  require_device(x, "input");
  const c10::DeviceGuard dg(x.device());
  TORCH_CHECK(x.dim() == 4 && w2.dim() == 4, "lokr_conv2d: NCHW input, w2 [c, d, kh, kw]");
  const int64_t B = x.size(0), C = x.size(1), H = x.size(2), W = x.size(3);
  const int64_t a = w1.size(0), b = w1.size(1), c = w2.size(0), d = w2.size(1);
  TORCH_CHECK(C == b * d, "adapter expects ", b * d, " input channels, got ", x.sizes());
  Geom gm = geom_of({w2.size(2), w2.size(3)}, stride, padding, dilation, H, W);
  bool copied;
  Tensor rows = rows_view(x, &copied), f1 = f32c(w1), w2p = f32c(w2.detach().permute({0, 2, 3, 1}));
  Tensor y = at::empty({B * gm.Ho * gm.Wo, a * c}, x.options());
  check_rc(lyc_lokr_conv2d_fwd(cptr(rows), cfp(f1), cfp(w2p), mptr(y), B, H, W, (int)a, (int)b, (int)c, (int)d, gm.kh, gm.kw, gm.sh,
                               gm.sw, gm.ph, gm.pw, gm.dh, gm.dw, (float)alpha, dtype_code(x.scalar_type()), stream_of(x)),
           "lyc_lokr_conv2d_fwd");
  return from_rows(y, B, gm.Ho, gm.Wo, !copied);
}

std::tuple<Tensor, Tensor, Tensor> lokr_conv2d_bwd(const Tensor& g, const Tensor& x, const Tensor& w1, const Tensor& w2, double alpha,
                                                   at::IntArrayRef stride, at::IntArrayRef padding, at::IntArrayRef dilation,
                                                   bool need_dx, bool need_dw1, bool need_dw2) {
  require_device(x, "input");
  const c10::DeviceGuard dg(x.device());
  const int64_t B = x.size(0), C = x.size(1), H = x.size(2), W = x.size(3);
  const int64_t a = w1.size(0), b = w1.size(1), c = w2.size(0), d = w2.size(1), kh = w2.size(2), kw = w2.size(3);
  bool xc, gc;
  Tensor rows = rows_view(x, &xc), g_rows = rows_view(g, &gc);
  Tensor f1 = f32c(w1), w2p = f32c(w2.detach().permute({0, 2, 3, 1}));
  const int code = dtype_code(x.scalar_type());
  const bool want_dx = need_dx || need_dw1;  // the w1 gradient shares the pass that produces dx
  Tensor dx_rows = want_dx ? at::empty({B * H * W, C}, rows.options()) : Tensor();
  Tensor dw1 = need_dw1 ? at::zeros({a, b}, w1.options().dtype(at::kFloat)) : Tensor();
  Tensor dw2p = need_dw2 ? at::zeros({c, kh, kw, d}, w2.options().dtype(at::kFloat)) : Tensor();
  Tensor ws, w2t;
  if (dw1.defined()) {
    const int64_t nb = lyc_lokr_conv2d_bwd_workspace_bytes(B, H, W, (int)a, (int)b, (int)d);
    if (nb > 0) ws = at::empty({nb}, rows.options().dtype(at::kByte));
  }
  if (dx_rows.defined() && stride[0] == 1 && stride[1] == 1) w2t = w2.detach().to(at::kFloat).permute({2, 3, 0, 1}).contiguous();
  check_rc(lyc_lokr_conv2d_bwd(cptr(g_rows), cptr(rows), cfp(f1), cfp(w2p), cfp(w2t), mptr(dx_rows), mfp(dw1), mfp(dw2p), mptr(ws), B,
                               H, W, (int)a, (int)b, (int)c, (int)d, (int)kh, (int)kw, (int)stride[0], (int)stride[1],
                               (int)padding[0], (int)padding[1], (int)dilation[0], (int)dilation[1], (float)alpha, code, stream_of(x)),
           "lyc_lokr_conv2d_bwd");
  Tensor dx = need_dx ? from_rows(dx_rows, B, H, W, !xc) : at::empty({0}, x.options());
  Tensor dw2;
  if (need_dw2) {  // in w2's own layout (what the Meta kernel promises: empty_like)
    dw2 = at::empty_like(w2);
    dw2.copy_(dw2p.permute({0, 3, 1, 2}));
  }
  return {dx, need_dw1 ? dw1.to(w1.scalar_type()) : at::empty({0}, w1.options()), need_dw2 ? dw2 : at::empty({0}, w2.options())};
}
std::tuple<Tensor, Tensor, Tensor> lokr_conv2d_bwd_meta(const Tensor& g, const Tensor& x, const Tensor& w1, const Tensor& w2,
                                                        double alpha, at::IntArrayRef, at::IntArrayRef, at::IntArrayRef, bool nx,
                                                        bool n1, bool n2) {
  return {nx ? at::empty_like(x) : x.new_empty({0}), n1 ? at::empty_like(w1) : w1.new_empty({0}),
          n2 ? at::empty_like(w2) : w2.new_empty({0})};
}

std::tuple<Tensor, Tensor> locon_conv2d_fwd(const Tensor& x, const Tensor& down, const Tensor& up, double alpha,
                                            at::IntArrayRef stride, at::IntArrayRef padding, at::IntArrayRef dilation) {
  require_device(x, "input");
  const c10::DeviceGuard dg(x.device());
  TORCH_CHECK(x.dim() == 4 && down.dim() == 4, "locon_conv2d: NCHW input, lora_down [r, C, kh, kw]");
  const int64_t B = x.size(0), C = x.size(1), H = x.size(2), W = x.size(3);
  const int64_t r = down.size(0), O = up.size(0);
  TORCH_CHECK(C == down.size(1), "adapter expects ", down.size(1), " input channels, got ", x.sizes());
  Geom gm = geom_of({down.size(2), down.size(3)}, stride, padding, dilation, H, W);
  bool copied;
  Tensor rows = rows_view(x, &copied), down_p = f32c(down.detach().permute({0, 2, 3, 1})), up2 = f32c(up.detach().reshape({O, r}));
  Tensor t = at::empty({B * gm.Ho * gm.Wo, r}, x.options().dtype(at::kFloat));
  Tensor y = at::empty({B * gm.Ho * gm.Wo, O}, x.options());
  check_rc(lyc_locon_conv2d_fwd(cptr(rows), cfp(down_p), cfp(up2), mfp(t), mptr(y), B, H, W, (int)C, (int)O, (int)r, gm.kh, gm.kw,
                                gm.sh, gm.sw, gm.ph, gm.pw, gm.dh, gm.dw, (float)alpha, dtype_code(x.scalar_type()), stream_of(x)),
           "lyc_locon_conv2d_fwd");
  return {from_rows(y, B, gm.Ho, gm.Wo, !copied), t};
}
std::tuple<Tensor, Tensor> locon_conv2d_fwd_meta(const Tensor& x, const Tensor& down, const Tensor& up, double alpha,
                                                 at::IntArrayRef stride, at::IntArrayRef padding, at::IntArrayRef dilation) {
  Tensor y = conv2d_meta(x, down, up, alpha, stride, padding, dilation, false);
  return {y, x.new_empty({y.size(0) * y.size(2) * y.size(3), down.size(0)}, x.options().dtype(at::kFloat))};
}

std::tuple<Tensor, Tensor, Tensor> locon_conv2d_bwd(const Tensor& g, const Tensor& x, const Tensor& down, const Tensor& up,
                                                    const Tensor& t, double alpha, at::IntArrayRef stride, at::IntArrayRef padding,
                                                    at::IntArrayRef dilation, bool need_dx, bool need_dd, bool need_du) {
  require_device(x, "input");
  const c10::DeviceGuard dg(x.device());
  const int64_t B = x.size(0), C = x.size(1), H = x.size(2), W = x.size(3);
  const int64_t r = down.size(0), O = up.size(0), kh = down.size(2), kw = down.size(3);
  Geom gm = geom_of({kh, kw}, stride, padding, dilation, H, W);
  bool xc, gc;
  Tensor rows = rows_view(x, &xc), g_rows = rows_view(g, &gc);
  Tensor down_p = f32c(down.detach().permute({0, 2, 3, 1})), up2 = f32c(up.detach().reshape({O, r}));
  Tensor dt = at::empty({B * gm.Ho * gm.Wo, r}, rows.options().dtype(at::kFloat));
  Tensor dx_rows = need_dx ? at::empty({B * H * W, C}, rows.options()) : Tensor();
  Tensor ddp = need_dd ? at::zeros({r, kh, kw, C}, down.options().dtype(at::kFloat)) : Tensor();
  Tensor du = need_du ? at::zeros({O, r}, up.options().dtype(at::kFloat)) : Tensor();
  check_rc(lyc_locon_conv2d_bwd(cptr(g_rows), cptr(rows), cfp(down_p), cfp(up2), cfp(t), mfp(dt), mptr(dx_rows), mfp(ddp), mfp(du), B,
                                H, W, (int)C, (int)O, (int)r, (int)kh, (int)kw, (int)stride[0], (int)stride[1], (int)padding[0],
                                (int)padding[1], (int)dilation[0], (int)dilation[1], (float)alpha, dtype_code(x.scalar_type()),
                                stream_of(x)), "lyc_locon_conv2d_bwd");
  Tensor dx = need_dx ? from_rows(dx_rows, B, H, W, !xc) : at::empty({0}, x.options());
  Tensor dd, duo;
  if (need_dd) {
    dd = at::empty_like(down);
    dd.copy_(ddp.permute({0, 3, 1, 2}));
  }
  if (need_du) {
    duo = at::empty_like(up);
    duo.copy_(du.reshape(up.sizes()));
  }
  return {dx, need_dd ? dd : at::empty({0}, down.options()), need_du ? duo : at::empty({0}, up.options())};
}
std::tuple<Tensor, Tensor, Tensor> locon_conv2d_bwd_meta(const Tensor& g, const Tensor& x, const Tensor& down, const Tensor& up,
                                                         const Tensor& t, double alpha, at::IntArrayRef, at::IntArrayRef,
                                                         at::IntArrayRef, bool nx, bool nd, bool nu) {
  return {nx ? at::empty_like(x) : x.new_empty({0}), nd ? at::empty_like(down) : down.new_empty({0}),
          nu ? at::empty_like(up) : up.new_empty({0})};
}

// autograd for traced (non-eager) tensors: forward = the public op below autograd, backward = the functional backward op
struct LokrConv2dTraceFn : public torch::autograd::Function<LokrConv2dTraceFn> {
  static Tensor forward(AutogradContext* ctx, const Tensor& x, const Tensor& w1, const Tensor& w2, double alpha,
                        std::vector<int64_t> stride, std::vector<int64_t> padding, std::vector<int64_t> dilation) {
    at::AutoDispatchBelowADInplaceOrView guard;
    static auto op = c10::Dispatcher::singleton().findSchemaOrThrow("lycoris_amd::lokr_conv2d", "")
                         .typed<Tensor(const Tensor&, const Tensor&, const Tensor&, double, at::IntArrayRef, at::IntArrayRef, at::IntArrayRef)>();
    Tensor y = op.call(x, w1, w2, alpha, stride, padding, dilation);
    save_vars(ctx, {x, w1, w2});
    ctx->saved_data["alpha"] = alpha;
    ctx->saved_data["stride"] = stride;
    ctx->saved_data["padding"] = padding;
    ctx->saved_data["dilation"] = dilation;
    return y;
  }
  static variable_list backward(AutogradContext* ctx, variable_list grads) {
    planes_mark_dirty_after_backward();  // the end of this backward pass is a step boundary for the plane cache (every LoKr node, on every path)
    auto s = saved_vars(ctx);
    const bool nx = ctx->needs_input_grad(0), n1 = ctx->needs_input_grad(1), n2 = ctx->needs_input_grad(2);
    static auto op = c10::Dispatcher::singleton().findSchemaOrThrow("lycoris_amd::_lokr_conv2d_backward", "")
                         .typed<std::tuple<Tensor, Tensor, Tensor>(const Tensor&, const Tensor&, const Tensor&, const Tensor&, double,
                                                                   at::IntArrayRef, at::IntArrayRef, at::IntArrayRef, bool, bool, bool)>();
    auto [dx, d1, d2] = op.call(grads[0], s[0], s[1], s[2], ctx->saved_data["alpha"].toDouble(), ctx->saved_data["stride"].toIntVector(),
                                ctx->saved_data["padding"].toIntVector(), ctx->saved_data["dilation"].toIntVector(), nx, n1, n2);
    return {nx ? dx : Tensor(), n1 ? d1 : Tensor(), n2 ? d2 : Tensor(), Tensor(), Tensor(), Tensor(), Tensor()};
  }
};
struct LoconConv2dTraceFn : public torch::autograd::Function<LoconConv2dTraceFn> {
  static Tensor forward(AutogradContext* ctx, const Tensor& x, const Tensor& down, const Tensor& up, double alpha,
                        std::vector<int64_t> stride, std::vector<int64_t> padding, std::vector<int64_t> dilation) {
    at::AutoDispatchBelowADInplaceOrView guard;
    static auto op = c10::Dispatcher::singleton().findSchemaOrThrow("lycoris_amd::_locon_conv2d_forward", "")
                         .typed<std::tuple<Tensor, Tensor>(const Tensor&, const Tensor&, const Tensor&, double, at::IntArrayRef,
                                                           at::IntArrayRef, at::IntArrayRef)>();
    auto [y, t] = op.call(x, down, up, alpha, stride, padding, dilation);
    save_vars(ctx, {x, down, up, t});
    ctx->saved_data["alpha"] = alpha;
    ctx->saved_data["stride"] = stride;
    ctx->saved_data["padding"] = padding;
    ctx->saved_data["dilation"] = dilation;
    return y;
  }
  static variable_list backward(AutogradContext* ctx, variable_list grads) {
    auto s = saved_vars(ctx);
    const bool nx = ctx->needs_input_grad(0), nd = ctx->needs_input_grad(1), nu = ctx->needs_input_grad(2);
    static auto op = c10::Dispatcher::singleton().findSchemaOrThrow("lycoris_amd::_locon_conv2d_backward", "")
                         .typed<std::tuple<Tensor, Tensor, Tensor>(const Tensor&, const Tensor&, const Tensor&, const Tensor&, const Tensor&,
                                                                   double, at::IntArrayRef, at::IntArrayRef, at::IntArrayRef, bool, bool,
                                                                   bool)>();
    auto [dx, dd, du] = op.call(grads[0], s[0], s[1], s[2], s[3], ctx->saved_data["alpha"].toDouble(),
                                ctx->saved_data["stride"].toIntVector(), ctx->saved_data["padding"].toIntVector(),
                                ctx->saved_data["dilation"].toIntVector(), nx, nd, nu);
    return {nx ? dx : Tensor(), nd ? dd : Tensor(), nu ? du : Tensor(), Tensor(), Tensor(), Tensor(), Tensor()};
  }
};

struct LokrConv2dFn : public torch::autograd::Function<LokrConv2dFn> {
  static Tensor forward(AutogradContext* ctx, const Tensor& x, const Tensor& w1, const Tensor& w2, double alpha,
                        std::vector<int64_t> stride, std::vector<int64_t> padding, std::vector<int64_t> dilation) {
    at::AutoDispatchBelowADInplaceOrView guard;
    require_device(x, "input");
    TORCH_CHECK(eager_cuda(x), "lycoris_amd::lokr_conv2d is an eager op (its forward / backward are not split into traceable ops yet)");
    const c10::DeviceGuard dg(x.device());
    TORCH_CHECK(x.dim() == 4 && w2.dim() == 4, "lokr_conv2d: NCHW input, w2 [c, d, kh, kw]");
    const int64_t B = x.size(0), C = x.size(1), H = x.size(2), W = x.size(3);
    const int64_t a = w1.size(0), b = w1.size(1), c = w2.size(0), d = w2.size(1);
    TORCH_CHECK(C == b * d, "adapter expects ", b * d, " input channels, got ", x.sizes());
    Geom gm = geom_of({w2.size(2), w2.size(3)}, stride, padding, dilation, H, W);
    bool copied;
    Tensor rows = rows_view(x, &copied), f1 = f32c(w1);
    Tensor y = at::empty({B * gm.Ho * gm.Wo, a * c}, x.options());
    const int code = dtype_code(x.scalar_type());
    // Pre-packed operand planes + LDS source patch (csrc/kron_conv.h) where the geometry allows: ONE small pack launch writes
    // w2 as hi / lo planes for the forward contraction and for the transposed one of the backward pass (kept for it), straight
    // from the parameter's own memory layout -- no permuted fp32 copies of w2.
    const bool pf = lyc_lokr_conv2d_planes_ok(B, H, W, (int)a, (int)b, (int)c, (int)d, gm.kh, gm.kw, gm.sh, gm.sw, gm.ph, gm.pw, gm.dh,
                                              gm.dw, code, 0) != 0;
    const bool pb = tl_grad_at_apply && (x.requires_grad() || w1.requires_grad() || w2.requires_grad()) &&
                    lyc_lokr_conv2d_planes_ok(B, H, W, (int)a, (int)b, (int)c, (int)d, gm.kh, gm.kw, gm.sh, gm.sw, gm.ph, gm.pw, gm.dh,
                                              gm.dw, code, 1) != 0;
    Tensor planes_f, planes_b;
    const int64_t bytes_f = lyc_lokr_planes_bytes((int)c, (int)d, gm.kh * gm.kw, 0), bytes_b = lyc_lokr_planes_bytes((int)c, (int)d, gm.kh * gm.kw, 1);
    Tensor cached = (pf || pb) ? planes_for(w2, x.scalar_type(), stream_of(x), /*fwd_role=*/true) : Tensor();  // leaf parameter: packed once per optimizer step
    if (cached.defined()) {
      if (pf) planes_f = cached.narrow(0, 0, bytes_f);
      if (pb) planes_b = cached.narrow(0, bytes_f, bytes_b);
    } else if (pf || pb) {
      Tensor w2f = w2.detach();
      if (w2f.scalar_type() != at::kFloat) w2f = w2f.to(at::kFloat);
      if (w2f.stride(2) != gm.kw * w2f.stride(3) && gm.kh > 1) w2f = w2f.contiguous();  // one tap stride: (i, j) -> i * kw + j
      const int taps = gm.kh * gm.kw;
      if (pf) planes_f = at::empty({lyc_lokr_planes_bytes((int)c, (int)d, taps, 0)}, x.options().dtype(at::kByte));
      if (pb) planes_b = at::empty({lyc_lokr_planes_bytes((int)c, (int)d, taps, 1)}, x.options().dtype(at::kByte));
      check_rc(lyc_lokr_pack_w2(cfp(w2f), w2f.stride(0), w2f.stride(1), w2f.stride(3), nullptr, 0, 0, nullptr, 0, 0, 0, 0, (int)c, (int)d,
                                taps, mptr(planes_f), mptr(planes_b), code, stream_of(x)), "lyc_lokr_pack_w2");
    }
    if (pf) {
      check_rc(lyc_lokr_conv2d_fwd_planes(cptr(rows), cfp(f1), cptr(planes_f), mptr(y), B, H, W, (int)a, (int)b, (int)c, (int)d, gm.kh,
                                          gm.kw, gm.sh, gm.sw, gm.ph, gm.pw, gm.dh, gm.dw, (float)alpha, code, stream_of(x)),
               "lyc_lokr_conv2d_fwd_planes");
    } else {
      Tensor w2p = f32c(w2.detach().permute({0, 2, 3, 1}));
      check_rc(lyc_lokr_conv2d_fwd(cptr(rows), cfp(f1), cfp(w2p), mptr(y), B, H, W, (int)a, (int)b, (int)c, (int)d, gm.kh, gm.kw, gm.sh,
                                   gm.sw, gm.ph, gm.pw, gm.dh, gm.dw, (float)alpha, code, stream_of(x)),
               "lyc_lokr_conv2d_fwd");
    }
    expect(w1, x);
    expect(w2, x, /*cl=*/true);
    save_vars(ctx, {rows, w1, w2, planes_b});
    ctx->saved_data["alpha"] = alpha;
    ctx->saved_data["geom"] = std::vector<int64_t>{gm.kh, gm.kw, gm.sh, gm.sw, gm.ph, gm.pw, gm.dh, gm.dw, gm.Ho, gm.Wo, B, C, H, W, !copied};
    return from_rows(y, B, gm.Ho, gm.Wo, !copied);
  }
  static variable_list backward(AutogradContext* ctx, variable_list grads) {
    planes_mark_dirty_after_backward();  // the end of this backward pass is a step boundary for the plane cache (every LoKr node, on every path)
    auto s = saved_vars(ctx);
    const Tensor &rows = s[0], &w1 = s[1], &w2 = s[2];
    const double alpha = ctx->saved_data["alpha"].toDouble();
    auto gv = ctx->saved_data["geom"].toIntVector();
    const int64_t B = gv[10], C = gv[11], H = gv[12], W = gv[13];
    const bool x_cl = gv[14] != 0;
    const int64_t a = w1.size(0), b = w1.size(1), c = w2.size(0), d = w2.size(1);
    const c10::DeviceGuard dg(rows.device());
    bool cp;
    Tensor g_rows = rows_view(grads[0], &cp);
    const bool nx = ctx->needs_input_grad(0), n1 = ctx->needs_input_grad(1), n2 = ctx->needs_input_grad(2);
    const Tensor& planes_b = s[3];  // defined: the transposed convolution runs the patch kernel on pre-packed planes
    Tensor f1 = f32c(w1), w2p;
    if (!planes_b.defined()) w2p = f32c(w2.detach().permute({0, 2, 3, 1}));
    const int code = dtype_code(rows.scalar_type());
    Tensor dx_rows = (nx || n1) ? at::empty({B * H * W, C}, rows.options()) : Tensor();
    const bool a1 = n1, a2 = n2;
    GradTarget t1 = grad_target(w1, a1);
    GradTarget t2 = cl_grad_target(w2, a2, {c, gv[0], gv[1], d});
    Tensor ws;
    if (t1.buf.defined()) {
      const int64_t nb = lyc_lokr_conv2d_bwd_workspace_bytes(B, H, W, (int)a, (int)b, (int)d);
      if (nb > 0) ws = at::empty({nb}, rows.options().dtype(at::kByte));
    }
    // both factor gradients go straight into .grad (fused accumulation): only the dx launch runs now, the weight gradients of
    // all parked Conv2d layers run in grouped launches at the end of the backward pass (as for nn.Linear)
    const bool defer = g_defer.enabled && planes_b.defined() && t2.buf.defined() && !t2.hand_back && t1.buf.defined() && !t1.hand_back &&
                       ws.defined() && dx_rows.defined();
    if (defer) {
      check_rc(lyc_lokr_conv2d_bwd_planes(cptr(g_rows), cptr(rows), cfp(f1), nullptr, cptr(planes_b), mptr(dx_rows), mfp(t1.buf),
                                          nullptr, mptr(ws), B, H, W, (int)a, (int)b, (int)c, (int)d, (int)gv[0], (int)gv[1],
                                          (int)gv[2], (int)gv[3], (int)gv[4], (int)gv[5], (int)gv[6], (int)gv[7], (float)alpha,
                                          code | LYC_DEFER_WGRAD, stream_of(rows)), "lyc_lokr_conv2d_bwd_planes(dx)");
      DeferredLokrConv item{g_rows, rows, f1, w1, w2, t1.buf, t2.buf, ws, B, H, W,
                            lyc_lokr_conv2d_dx_blocks(B, H, W, (int)a, (int)b, (int)c, (int)d, (int)gv[0], (int)gv[1], (int)gv[2], (int)gv[3],
                                                      (int)gv[4], (int)gv[5], (int)gv[6], (int)gv[7], code, 1),
                            (int)a, (int)b, (int)c, (int)d, {(int)gv[0], (int)gv[1], (int)gv[2], (int)gv[3], (int)gv[4], (int)gv[5], (int)gv[6],
                            (int)gv[7]}, code, (float)alpha, stream_of(rows), rows.device().index()};
      park_deferred(std::move(item));
      Tensor dxd = nx ? from_rows(dx_rows, B, H, W, x_cl) : Tensor();
      return {dxd, Tensor(), Tensor(), Tensor(), Tensor(), Tensor(), Tensor()};
    }
    if (planes_b.defined()) {
      check_rc(lyc_lokr_conv2d_bwd_planes(cptr(g_rows), cptr(rows), cfp(f1), nullptr, cptr(planes_b), mptr(dx_rows), mfp(t1.buf),
                                          mfp(t2.buf), mptr(ws), B, H, W, (int)a, (int)b, (int)c, (int)d, (int)gv[0], (int)gv[1],
                                          (int)gv[2], (int)gv[3], (int)gv[4], (int)gv[5], (int)gv[6], (int)gv[7], (float)alpha, code,
                                          stream_of(rows)), "lyc_lokr_conv2d_bwd_planes");
    } else {
      Tensor w2t;  // stride 1: a [kh, kw, c, d] copy lets the transposed convolution use full K segments
      if (dx_rows.defined() && gv[2] == 1 && gv[3] == 1) w2t = w2.detach().to(at::kFloat).permute({2, 3, 0, 1}).contiguous();
      check_rc(lyc_lokr_conv2d_bwd(cptr(g_rows), cptr(rows), cfp(f1), cfp(w2p), cfp(w2t), mptr(dx_rows), mfp(t1.buf), mfp(t2.buf),
                                   mptr(ws), B, H, W, (int)a, (int)b, (int)c, (int)d, (int)gv[0], (int)gv[1], (int)gv[2], (int)gv[3],
                                   (int)gv[4], (int)gv[5], (int)gv[6], (int)gv[7], (float)alpha, code, stream_of(rows)),
               "lyc_lokr_conv2d_bwd");
    }
    Tensor dx = nx ? from_rows(dx_rows, B, H, W, x_cl) : Tensor();
    return {dx, finish_grad(w1, t1), cl_finish(w2, t2), Tensor(), Tensor(), Tensor(), Tensor()};
  }
};
Tensor lokr_conv2d_implicit(const Tensor& x, const Tensor& w1, const Tensor& w2, double alpha, at::IntArrayRef stride,
                            at::IntArrayRef padding, at::IntArrayRef dilation) {
  if (!x.is_cuda() || !eager_cuda(x))  // FakeTensor / meta: differentiable through the functional forward / backward ops
    return LokrConv2dTraceFn::apply(amp(x), w1, w2, alpha, stride.vec(), padding.vec(), dilation.vec());
  const GradAtApply ga_;
  return LokrConv2dFn::apply(amp(x), w1, w2, alpha, stride.vec(), padding.vec(), dilation.vec());
}

// ---- LoKr on nn.Conv2d with a low-rank w2 = w2a [c, r] @ w2b [r, d*kh*kw] (reference modules/lokr.py:131-136, 370) -----------------
// Only where the patch kernels take both passes (lokr_conv2d_lr_ok): the planes are packed straight from the two factors, the
// weight gradient goes through the grouped chain-rule launch.  Everything else forms the product (lycoris_amd/ops.py).
bool lokr_conv2d_lr_ok(const Tensor& x, const Tensor& w1, const Tensor& w2a, const Tensor& w2b, at::IntArrayRef kernel,
                       at::IntArrayRef stride, at::IntArrayRef padding, at::IntArrayRef dilation) {
  if (!x.is_cuda() || !eager_cuda(x) || x.dim() != 4 || w1.dim() != 2 || w2a.dim() != 2 || w2b.dim() != 2) return false;
  at::ScalarType act = x.scalar_type();
  if (act == at::kFloat && at::autocast::is_autocast_enabled(at::kCUDA)) act = at::autocast::get_autocast_dtype(at::kCUDA);
  if (act != at::kBFloat16 && act != at::kHalf) return false;
  const int taps = (int)(kernel[0] * kernel[1]);
  if (taps < 1 || w2b.size(1) % taps || w2a.size(1) != w2b.size(0)) return false;
  const int64_t a = w1.size(0), b = w1.size(1), c = w2a.size(0), d = w2b.size(1) / taps;
  if (x.size(1) != b * d || (c % 8) || (d % 8)) return false;
  if (!g_planes.enabled || !w2a.is_leaf() || !w2b.is_leaf() || w2a.scalar_type() != at::kFloat || w2b.scalar_type() != at::kFloat ||
      !w2a.is_contiguous() || !w2b.is_contiguous())
    return false;
  Geom gm = geom_of(kernel, stride, padding, dilation, x.size(2), x.size(3));
  const int code = dtype_code(act);
  for (int bw = 0; bw < 2; ++bw)
    if (!lyc_lokr_conv2d_planes_ok(x.size(0), x.size(2), x.size(3), (int)a, (int)b, (int)c, (int)d, gm.kh, gm.kw, gm.sh, gm.sw, gm.ph, gm.pw,
                                   gm.dh, gm.dw, code, bw))
      return false;
  return true;
}

struct LokrConv2dLrFn : public torch::autograd::Function<LokrConv2dLrFn> {
  static Tensor forward(AutogradContext* ctx, const Tensor& x, const Tensor& w1, const Tensor& w2a, const Tensor& w2b, double alpha,
                        std::vector<int64_t> kernel, std::vector<int64_t> stride, std::vector<int64_t> padding, std::vector<int64_t> dilation) {
    at::AutoDispatchBelowADInplaceOrView guard;
    require_device(x, "input");
    TORCH_CHECK(lokr_conv2d_lr_ok(x, w1, w2a, w2b, kernel, stride, padding, dilation),
                "lycoris_amd::lokr_conv2d_lr: layer off the patch kernels' fast path (form w2a @ w2b and call lokr_conv2d: ops.lokr_conv2d_lr does)");
    const c10::DeviceGuard dg(x.device());
    const int taps = (int)(kernel[0] * kernel[1]);
    const int64_t B = x.size(0), C = x.size(1), H = x.size(2), W = x.size(3);
    const int64_t a = w1.size(0), b = w1.size(1), c = w2a.size(0), d = w2b.size(1) / taps;
    Geom gm = geom_of(kernel, stride, padding, dilation, H, W);
    bool copied;
    Tensor rows = rows_view(x, &copied), f1 = f32c(w1);
    Tensor y = at::empty({B * gm.Ho * gm.Wo, a * c}, x.options());
    const int code = dtype_code(x.scalar_type());
    Tensor planes = planes_for_lr(w2a, w2b, x.scalar_type(), stream_of(x), taps, /*fwd_role=*/true);
    TORCH_CHECK(planes.defined(), "lokr_conv2d_lr: no operand planes");
    check_rc(lyc_lokr_conv2d_fwd_planes(cptr(rows), cfp(f1), cptr(planes), mptr(y), B, H, W, (int)a, (int)b, (int)c, (int)d, gm.kh, gm.kw,
                                        gm.sh, gm.sw, gm.ph, gm.pw, gm.dh, gm.dw, (float)alpha, code, stream_of(x)),
             "lyc_lokr_conv2d_fwd_planes");
    expect(w1, x);
    expect(w2a, x);
    expect(w2b, x);
    save_vars(ctx, {rows, w1, w2a, w2b});
    ctx->saved_data["alpha"] = alpha;
    ctx->saved_data["geom"] = std::vector<int64_t>{gm.kh, gm.kw, gm.sh, gm.sw, gm.ph, gm.pw, gm.dh, gm.dw, gm.Ho, gm.Wo, B, C, H, W, !copied};
    return from_rows(y, B, gm.Ho, gm.Wo, !copied);
  }
  static variable_list backward(AutogradContext* ctx, variable_list grads) {
    planes_mark_dirty_after_backward();  // the end of this backward pass is a step boundary for the plane cache (every LoKr node, on every path)
    auto s = saved_vars(ctx);
    const Tensor &rows = s[0], &w1 = s[1], &w2a = s[2], &w2b = s[3];
    const double alpha = ctx->saved_data["alpha"].toDouble();
    auto gv = ctx->saved_data["geom"].toIntVector();
    const int64_t B = gv[10], C = gv[11], H = gv[12], W = gv[13];
    const bool x_cl = gv[14] != 0;
    const int taps = (int)(gv[0] * gv[1]);
    const int64_t a = w1.size(0), b = w1.size(1), c = w2a.size(0), r = w2a.size(1), d = w2b.size(1) / taps;
    const c10::DeviceGuard dg(rows.device());
    bool cp;
    Tensor g_rows = rows_view(grads[0], &cp);
    const bool nx = ctx->needs_input_grad(0), n1 = ctx->needs_input_grad(1), na = ctx->needs_input_grad(2), nb2 = ctx->needs_input_grad(3);
    Tensor f1 = f32c(w1);
    const int code = dtype_code(rows.scalar_type());
    const bool a1 = n1, want_w2 = na || nb2;
    Tensor dx_rows = (nx || a1) ? at::empty({B * H * W, C}, rows.options()) : Tensor();
    GradTarget t1 = grad_target(w1, a1), ta = grad_target(w2a, want_w2), tb = grad_target(w2b, want_w2);
    Tensor ws;
    if (t1.buf.defined()) {
      const int64_t nb = lyc_lokr_conv2d_bwd_workspace_bytes(B, H, W, (int)a, (int)b, (int)d);
      if (nb > 0) ws = at::empty({nb}, rows.options().dtype(at::kByte));
    }
    // the planes of THIS step (the optimizer has not run between forward and backward: the cache entry is current)
    Tensor planes = planes_for_lr(w2a, w2b, rows.scalar_type(), stream_of(rows), taps);
    TORCH_CHECK(planes.defined(), "lokr_conv2d_lr: no operand planes in backward");
    const void* planes_b = planes_bwd_ptr(planes, c, d, taps);
    const bool defer = g_defer.enabled && want_w2 && !ta.hand_back && !tb.hand_back && t1.buf.defined() && !t1.hand_back && ws.defined() &&
                       dx_rows.defined();
    if (defer) {
      check_rc(lyc_lokr_conv2d_bwd_planes(cptr(g_rows), cptr(rows), cfp(f1), nullptr, planes_b, mptr(dx_rows), mfp(t1.buf), nullptr, mptr(ws),
                                          B, H, W, (int)a, (int)b, (int)c, (int)d, (int)gv[0], (int)gv[1], (int)gv[2], (int)gv[3], (int)gv[4],
                                          (int)gv[5], (int)gv[6], (int)gv[7], (float)alpha, code | LYC_DEFER_WGRAD, stream_of(rows)),
               "lyc_lokr_conv2d_bwd_planes(dx)");
      DeferredLokrConv item{g_rows, rows, f1, w1, Tensor(), t1.buf, Tensor(), ws, B, H, W,
                            lyc_lokr_conv2d_dx_blocks(B, H, W, (int)a, (int)b, (int)c, (int)d, (int)gv[0], (int)gv[1], (int)gv[2], (int)gv[3],
                                                      (int)gv[4], (int)gv[5], (int)gv[6], (int)gv[7], code, 1),
                            (int)a, (int)b, (int)c, (int)d, {(int)gv[0], (int)gv[1], (int)gv[2], (int)gv[3], (int)gv[4], (int)gv[5], (int)gv[6],
                            (int)gv[7]}, code, (float)alpha, stream_of(rows), rows.device().index()};
      item.w2a = w2a; item.w2b = w2b; item.d_w2a = ta.buf; item.d_w2b = tb.buf;
      park_deferred(std::move(item));
      return {nx ? from_rows(dx_rows, B, H, W, x_cl) : Tensor(), Tensor(), Tensor(), Tensor(), Tensor(), Tensor(), Tensor(), Tensor(), Tensor()};
    }
    Tensor dw2p = want_w2 ? at::zeros({c, gv[0], gv[1], d}, rows.options().dtype(at::kFloat)) : Tensor();
    check_rc(lyc_lokr_conv2d_bwd_planes(cptr(g_rows), cptr(rows), cfp(f1), nullptr, planes_b, mptr(dx_rows), mfp(t1.buf), mfp(dw2p), mptr(ws),
                                        B, H, W, (int)a, (int)b, (int)c, (int)d, (int)gv[0], (int)gv[1], (int)gv[2], (int)gv[3], (int)gv[4],
                                        (int)gv[5], (int)gv[6], (int)gv[7], (float)alpha, code, stream_of(rows)),
             "lyc_lokr_conv2d_bwd_planes");
    if (want_w2) {
      LycLokrLrChainItem ci{cfp(dw2p), cfp(w2a), cfp(w2b), mfp(ta.buf), mfp(tb.buf), (int)c, (int)d, (int)r, taps};
      check_rc(lyc_lokr_lr_chain_group(&ci, 1, stream_of(rows)), "lyc_lokr_lr_chain_group");
    }
    return {nx ? from_rows(dx_rows, B, H, W, x_cl) : Tensor(), finish_grad(w1, t1), finish_grad(w2a, ta), finish_grad(w2b, tb), Tensor(),
            Tensor(), Tensor(), Tensor(), Tensor()};
  }
};
// below autograd (inference_mode): the forward launch alone
Tensor lokr_conv2d_lr_cuda(const Tensor& x, const Tensor& w1, const Tensor& w2a, const Tensor& w2b, double alpha, at::IntArrayRef kernel,
                           at::IntArrayRef stride, at::IntArrayRef padding, at::IntArrayRef dilation) {
  TORCH_CHECK(lokr_conv2d_lr_ok(x, w1, w2a, w2b, kernel, stride, padding, dilation), "lycoris_amd::lokr_conv2d_lr: layer off the patch kernels' fast path");
  const c10::DeviceGuard dg(x.device());
  const int taps = (int)(kernel[0] * kernel[1]);
  const int64_t B = x.size(0), H = x.size(2), W = x.size(3), a = w1.size(0), b = w1.size(1), c = w2a.size(0), d = w2b.size(1) / taps;
  Geom gm = geom_of(kernel, stride, padding, dilation, H, W);
  bool copied;
  Tensor rows = rows_view(x, &copied), f1 = f32c(w1);
  Tensor y = at::empty({B * gm.Ho * gm.Wo, a * c}, x.options());
  Tensor planes = planes_for_lr(w2a, w2b, x.scalar_type(), stream_of(x), taps, /*fwd_role=*/true);
  TORCH_CHECK(planes.defined(), "lokr_conv2d_lr: no operand planes");
  check_rc(lyc_lokr_conv2d_fwd_planes(cptr(rows), cfp(f1), cptr(planes), mptr(y), B, H, W, (int)a, (int)b, (int)c, (int)d, gm.kh, gm.kw, gm.sh,
                                      gm.sw, gm.ph, gm.pw, gm.dh, gm.dw, (float)alpha, dtype_code(x.scalar_type()), stream_of(x)),
           "lyc_lokr_conv2d_fwd_planes");
  return from_rows(y, B, gm.Ho, gm.Wo, !copied);
}
Tensor lokr_conv2d_lr_meta(const Tensor& x, const Tensor& w1, const Tensor& w2a, const Tensor& w2b, double alpha, at::IntArrayRef kernel,
                           at::IntArrayRef stride, at::IntArrayRef padding, at::IntArrayRef dilation) {
  Geom gm = geom_of(kernel, stride, padding, dilation, x.size(2), x.size(3));
  return at::empty({x.size(0), w1.size(0) * w2a.size(0), gm.Ho, gm.Wo},
                   x.options().memory_format(rows_are_free(x) ? at::MemoryFormat::ChannelsLast : at::MemoryFormat::Contiguous));
}
Tensor lokr_conv2d_lr_autograd(const Tensor& x, const Tensor& w1, const Tensor& w2a, const Tensor& w2b, double alpha, at::IntArrayRef kernel,
                               at::IntArrayRef stride, at::IntArrayRef padding, at::IntArrayRef dilation) {
  if (!eager_cuda(x)) {  // tracing: the product as a [c, d, kh, kw] tensor through the traceable full-matrix Conv2d op
    static auto op = c10::Dispatcher::singleton().findSchemaOrThrow("lycoris_amd::lokr_conv2d", "")
                         .typed<Tensor(const Tensor&, const Tensor&, const Tensor&, double, at::IntArrayRef, at::IntArrayRef, at::IntArrayRef)>();
    Tensor w2 = at::matmul(w2a, w2b).reshape({w2a.size(0), -1, kernel[0], kernel[1]});
    return op.call(x, w1, w2, alpha, stride, padding, dilation);
  }
  const GradAtApply ga_;
  return LokrConv2dLrFn::apply(amp(x), w1, w2a, w2b, alpha, kernel.vec(), stride.vec(), padding.vec(), dilation.vec());
}

struct LoconConv2dFn : public torch::autograd::Function<LoconConv2dFn> {
  static Tensor forward(AutogradContext* ctx, const Tensor& x, const Tensor& down, const Tensor& up, double alpha,
                        std::vector<int64_t> stride, std::vector<int64_t> padding, std::vector<int64_t> dilation) {
    at::AutoDispatchBelowADInplaceOrView guard;
    require_device(x, "input");
    TORCH_CHECK(eager_cuda(x), "lycoris_amd::locon_conv2d is an eager op (its forward / backward are not split into traceable ops yet)");
    const c10::DeviceGuard dg(x.device());
    TORCH_CHECK(x.dim() == 4 && down.dim() == 4, "locon_conv2d: NCHW input, lora_down [r, C, kh, kw]");
    const int64_t B = x.size(0), C = x.size(1), H = x.size(2), W = x.size(3);
    const int64_t r = down.size(0), O = up.size(0);
    TORCH_CHECK(C == down.size(1), "adapter expects ", down.size(1), " input channels, got ", x.sizes());
    Geom gm = geom_of({down.size(2), down.size(3)}, stride, padding, dilation, H, W);
    bool copied;
    Tensor rows = rows_view(x, &copied), down_p = f32c(down.detach().permute({0, 2, 3, 1})), up2 = f32c(up.detach().reshape({O, r}));
    Tensor t = at::empty({B * gm.Ho * gm.Wo, r}, x.options().dtype(at::kFloat));
    Tensor y = at::empty({B * gm.Ho * gm.Wo, O}, x.options());
    check_rc(lyc_locon_conv2d_fwd(cptr(rows), cfp(down_p), cfp(up2), mfp(t), mptr(y), B, H, W, (int)C, (int)O, (int)r, gm.kh, gm.kw,
                                  gm.sh, gm.sw, gm.ph, gm.pw, gm.dh, gm.dw, (float)alpha, dtype_code(x.scalar_type()), stream_of(x)),
             "lyc_locon_conv2d_fwd");
    expect(down, x, /*cl=*/true);
    expect(up, x);
    save_vars(ctx, {rows, down, up, t});
    ctx->saved_data["alpha"] = alpha;
    ctx->saved_data["geom"] = std::vector<int64_t>{gm.kh, gm.kw, gm.sh, gm.sw, gm.ph, gm.pw, gm.dh, gm.dw, gm.Ho, gm.Wo, B, C, H, W, !copied};
    return from_rows(y, B, gm.Ho, gm.Wo, !copied);
  }
  static variable_list backward(AutogradContext* ctx, variable_list grads) {
    auto s = saved_vars(ctx);
    const Tensor &rows = s[0], &down = s[1], &up = s[2], &t = s[3];
    const double alpha = ctx->saved_data["alpha"].toDouble();
    auto gv = ctx->saved_data["geom"].toIntVector();
    const int64_t B = gv[10], C = gv[11], H = gv[12], W = gv[13], Ho = gv[8], Wo = gv[9];
    const bool x_cl = gv[14] != 0;
    const int64_t r = down.size(0), O = up.size(0);
    const c10::DeviceGuard dg(rows.device());
    bool cp;
    Tensor g_rows = rows_view(grads[0], &cp);
    const bool nx = ctx->needs_input_grad(0), nd = ctx->needs_input_grad(1), nu = ctx->needs_input_grad(2);
    Tensor down_p = f32c(down.detach().permute({0, 2, 3, 1})), up2 = f32c(up.detach().reshape({O, r}));
    Tensor dt = at::empty({B * Ho * Wo, r}, rows.options().dtype(at::kFloat));
    Tensor dx_rows = nx ? at::empty({B * H * W, C}, rows.options()) : Tensor();
    GradTarget td = cl_grad_target(down, nd, {r, gv[0], gv[1], C});
    GradTarget tu = grad_target(up, nu);
    check_rc(lyc_locon_conv2d_bwd(cptr(g_rows), cptr(rows), cfp(down_p), cfp(up2), cfp(t), mfp(dt), mptr(dx_rows), mfp(td.buf),
                                  mfp(tu.buf), B, H, W, (int)C, (int)O, (int)r, (int)gv[0], (int)gv[1], (int)gv[2], (int)gv[3],
                                  (int)gv[4], (int)gv[5], (int)gv[6], (int)gv[7], (float)alpha, dtype_code(rows.scalar_type()),
                                  stream_of(rows)), "lyc_locon_conv2d_bwd");
    Tensor dx = nx ? from_rows(dx_rows, B, H, W, x_cl) : Tensor();
    return {dx, cl_finish(down, td), finish_grad(up, tu), Tensor(), Tensor(), Tensor(), Tensor()};
  }
};
Tensor locon_conv2d_implicit(const Tensor& x, const Tensor& down, const Tensor& up, double alpha, at::IntArrayRef stride,
                             at::IntArrayRef padding, at::IntArrayRef dilation) {
  if (!x.is_cuda() || !eager_cuda(x))
    return LoconConv2dTraceFn::apply(amp(x), down, up, alpha, stride.vec(), padding.vec(), dilation.vec());
  const GradAtApply ga_;
  return LoconConv2dFn::apply(amp(x), down, up, alpha, stride.vec(), padding.vec(), dilation.vec());
}

Tensor lokr_conv2d_eager(const Tensor& x, const Tensor& w1, const Tensor& w2, double alpha, at::IntArrayRef stride,
                         at::IntArrayRef padding, at::IntArrayRef dilation) {
  return lokr_conv2d_fwd(x, w1, w2, alpha, stride, padding, dilation);  // below autograd: inference_mode / a compiled graph
}
Tensor locon_conv2d_eager(const Tensor& x, const Tensor& down, const Tensor& up, double alpha, at::IntArrayRef stride,
                          at::IntArrayRef padding, at::IntArrayRef dilation) {
  return std::get<0>(locon_conv2d_fwd(x, down, up, alpha, stride, padding, dilation));
}

Tensor conv2d_meta(const Tensor& x, const Tensor& f0, const Tensor& f1, double alpha, at::IntArrayRef stride, at::IntArrayRef padding,
                   at::IntArrayRef dilation, bool lokr) {
  const int64_t kh = lokr ? f1.size(2) : f0.size(2), kw = lokr ? f1.size(3) : f0.size(3);
  const int64_t O = lokr ? f0.size(0) * f1.size(0) : f1.size(0);
  Geom gm = geom_of({kh, kw}, stride, padding, dilation, x.size(2), x.size(3));
  // the kernels return the row matrix as a channels_last tensor when x came as one (rows_view / from_rows)
  return at::empty({x.size(0), O, gm.Ho, gm.Wo},
                   x.options().memory_format(rows_are_free(x) ? at::MemoryFormat::ChannelsLast : at::MemoryFormat::Contiguous));
}
Tensor lokr_conv2d_meta(const Tensor& x, const Tensor& w1, const Tensor& w2, double alpha, at::IntArrayRef s, at::IntArrayRef p,
                        at::IntArrayRef d) {
  return conv2d_meta(x, w1, w2, alpha, s, p, d, true);
}
Tensor locon_conv2d_meta(const Tensor& x, const Tensor& down, const Tensor& up, double alpha, at::IntArrayRef s, at::IntArrayRef p,
                         at::IntArrayRef d) {
  return conv2d_meta(x, down, up, alpha, s, p, d, false);
}

// =====================================================================================================================
// Conv2d through the row kernels: every adapter on every Conv2d geometry the implicit kernels above do not take
// (LoHa always; LoKr / LoCon with fp32 activations or factor shapes off their fast paths; 1x1 convolutions)
// =====================================================================================================================
// The Conv2d form of an adapter is its Linear form on the im2col matrix (include/lycoris_amd.h "Conv2d lowering"; reference:
// F.conv2d in functional/general.py:6 with the [.., I*kh*kw] factor views of modules/{locon,loha,lokr}.py).  A 1x1 / stride 1 /
// no-padding convolution needs no im2col at all: its row matrix is the NHWC view of x (free for a channels_last tensor).
// algo: 0 = LoKr (f0 = w1 [a, b], f1 = w2 [c, d*kh*kw]), 1 = LoCon (f0 = down [r, C*kh*kw], f1 = up [O, r]),
//       2 = LoHa (f0..f3 = w1a [O, r], w1b [r, C*kh*kw], w2a, w2b)
enum { ALGO_LOKR = 0, ALGO_LOCON = 1, ALGO_LOHA = 2 };
bool pointwise(const Geom& g) { return g.kh == 1 && g.kw == 1 && g.sh == 1 && g.sw == 1 && g.ph == 0 && g.pw == 0; }
int64_t rows_out_features(int64_t algo, const Tensor& f0, const Tensor& f1) {
  return algo == ALGO_LOKR ? f0.size(0) * f1.size(0) : algo == ALGO_LOCON ? f1.size(0) : f0.size(0);
}
int64_t rows_in_features(int64_t algo, const Tensor& f0, const Tensor& f1) {
  return algo == ALGO_LOKR ? f0.size(1) * f1.size(1) : algo == ALGO_LOCON ? f0.size(1) : f1.size(1);
}
// LoHa on a 16-bit Conv2d with C % 8 == 0 (round 6): the im2col matrix is built from the NHWC row matrix with WINDOW-MAJOR columns
// (lyc_im2col_rows: 16-byte vectors, a channels_last tensor needs no copy at all) and meets the factors with their columns permuted the
// same way.  The NCHW form (lyc_im2col: an element-wise gather, 183 us per SDXL layer; lyc_col2im: 165 us) cost 13 of LoHa's 88 ms.
bool window_major_cols(const Tensor& x, int64_t algo, const Geom& gm, int64_t C) {
  return algo == 2 /* ALGO_LOHA */ && x.scalar_type() != at::kFloat && (C % 8) == 0 && !(gm.kh == 1 && gm.kw == 1 && gm.sh == 1 && gm.sw == 1 && gm.ph == 0 && gm.pw == 0);
}
// [r, C*kk] (channel-major, the reference's layout) -> [r, kk*C] (window-major), fp32, contiguous
Tensor to_window_major(const Tensor& f, int64_t C, int64_t kk) {
  const int64_t r = f.size(0);
  return f32c(f.detach()).view({r, C, kk}).transpose(1, 2).contiguous().view({r, kk * C});
}
Tensor im2col_rows_wm(const Tensor& x, const Geom& gm) {
  const int64_t B = x.size(0), C = x.size(1), H = x.size(2), W = x.size(3);
  bool copied;
  Tensor xr = rows_view(x, &copied);
  Tensor cols = at::empty({B * gm.Ho * gm.Wo, gm.kh * gm.kw * C}, x.options());
  check_rc(lyc_im2col_rows(cptr(xr), mptr(cols), B, C, H, W, gm.kh, gm.kw, gm.sh, gm.sw, gm.ph, gm.pw, gm.dh, gm.dw, dtype_code(x.scalar_type()),
                           stream_of(x)), "lyc_im2col_rows");
  return cols;
}
Tensor im2col_rows(const Tensor& x, const Geom& gm) {
  const int64_t B = x.size(0), C = x.size(1), H = x.size(2), W = x.size(3);
  Tensor xc = x.contiguous();
  Tensor cols = at::empty({B * gm.Ho * gm.Wo, C * gm.kh * gm.kw}, x.options());
  check_rc(lyc_im2col(cptr(xc), mptr(cols), B, C, H, W, gm.kh, gm.kw, gm.sh, gm.sw, gm.ph, gm.pw, gm.dh, gm.dw, dtype_code(x.scalar_type()),
                      stream_of(x)), "lyc_im2col");
  return cols;
}
// (y [B, O, Ho, Wo], cols: the im2col matrix (empty for a 1x1 convolution: backward takes the rows from x), saved: LoCon's t /
// LoHa's dW images)
std::tuple<Tensor, Tensor, Tensor> adapter_conv2d_fwd(const Tensor& x, const Tensor& f0, const Tensor& f1, const c10::optional<Tensor>& f2,
                                                      const c10::optional<Tensor>& f3, int64_t algo, double alpha, at::IntArrayRef kernel,
                                                      at::IntArrayRef stride, at::IntArrayRef padding, at::IntArrayRef dilation) {
  require_device(x, "input");
  const c10::DeviceGuard dg(x.device());
  TORCH_CHECK(x.dim() == 4, "Conv2d adapter expects NCHW input, got shape ", x.sizes());
  TORCH_CHECK(algo >= ALGO_LOKR && algo <= ALGO_LOHA, "adapter_conv2d: algo ", algo);
  TORCH_CHECK(algo != ALGO_LOHA || (f2.has_value() && f3.has_value()), "adapter_conv2d: LoHa takes four factors");
  const int64_t B = x.size(0), C = x.size(1), H = x.size(2), W = x.size(3);
  Geom gm = geom_of(kernel, stride, padding, dilation, H, W);
  const int64_t I = rows_in_features(algo, f0, f1), O = rows_out_features(algo, f0, f1);
  TORCH_CHECK(I == C * gm.kh * gm.kw, "adapter expects ", I, " = C*kh*kw im2col features, input has C=", C, ", kernel=", kernel);
  const bool pw = pointwise(gm);
  const bool wm = window_major_cols(x, algo, gm, C);
  bool copied = true;
  Tensor rows = pw ? rows_view(x, &copied) : (wm ? im2col_rows_wm(x, gm) : im2col_rows(x, gm));
  Tensor y_rows, saved;
  if (algo == ALGO_LOKR) {
    y_rows = lokr_linear_fwd(rows, f0, f1, alpha, c10::nullopt);
    saved = at::empty({0}, x.options());
  } else if (algo == ALGO_LOCON) {
    std::tie(y_rows, saved) = locon_linear_fwd(rows, f0, f1, alpha);
  } else if (wm) {
    const int64_t kk = gm.kh * gm.kw;
    std::tie(y_rows, saved) = loha_linear_fwd(rows, f0, to_window_major(f1, C, kk), *f2, to_window_major(*f3, C, kk), alpha);
  } else {
    std::tie(y_rows, saved) = loha_linear_fwd(rows, f0, f1, *f2, *f3, alpha);
  }
  (void)O;
  // (window-major path: the layer output of a channels_last input is handed back channels_last -- a free view of the row matrix)
  return {from_rows(y_rows, B, gm.Ho, gm.Wo, (pw && !copied) || (wm && rows_are_free(x))), pw ? at::empty({0}, x.options()) : rows, saved};
}

// dx and the factor gradients accumulated into the defined d[i]
Tensor adapter_conv2d_bwd_into(const Tensor& g, const Tensor& x_or_cols, bool is_cols, at::IntArrayRef xshape, bool x_cl, const Tensor& saved,
                               const Tensor* f[4], int64_t algo, double alpha, const Geom& gm, bool need_dx, Tensor (&d)[4]) {
  const c10::DeviceGuard dg(g.device());
  const int64_t B = xshape[0], H = xshape[2], W = xshape[3];
  bool cp;
  Tensor g_rows = rows_view(g, &cp);
  Tensor rows = is_cols ? x_or_cols : rows_view(x_or_cols, &cp);
  const bool f32_rows = is_cols && g.scalar_type() != at::kFloat;  // col2im sums up to kh*kw row entries per pixel: fp32, rounded once
  Tensor dx_rows;
  const int64_t C = xshape[1];
  const bool wm = is_cols && window_major_cols(g, algo, gm, C);
  if (algo == ALGO_LOKR) dx_rows = lokr_linear_bwd_into(g_rows, rows, *f[0], *f[1], alpha, need_dx, d[0], d[1], f32_rows);
  else if (algo == ALGO_LOCON) dx_rows = locon_linear_bwd_into(g_rows, rows, *f[0], *f[1], saved, alpha, need_dx, d[0], d[1], f32_rows);
  else if (wm) {
    // window-major columns: the b-side factors and their gradients in the permuted layout; un-permuted into the callers' buffers
    const int64_t kk = gm.kh * gm.kw, r = f[1]->size(0);
    Tensor b1 = to_window_major(*f[1], C, kk), b2 = to_window_major(*f[3], C, kk);
    Tensor dw[4] = {d[0], Tensor(), d[2], Tensor()};
    if (d[1].defined()) dw[1] = at::zeros({r, kk * C}, b1.options());
    if (d[3].defined()) dw[3] = at::zeros({r, kk * C}, b2.options());
    dx_rows = loha_linear_bwd_into(g_rows, rows, *f[0], b1, *f[2], b2, saved, alpha, need_dx, dw, f32_rows);
    if (d[1].defined()) d[1].view({r, C, kk}).add_(dw[1].view({r, kk, C}).transpose(1, 2));
    if (d[3].defined()) d[3].view({r, C, kk}).add_(dw[3].view({r, kk, C}).transpose(1, 2));
  } else dx_rows = loha_linear_bwd_into(g_rows, rows, *f[0], *f[1], *f[2], *f[3], saved, alpha, need_dx, d, f32_rows);
  if (!need_dx) return Tensor();
  if (!is_cols) return from_rows(dx_rows, B, H, W, x_cl);
  if (wm) {
    Tensor dxr = at::empty({B * H * W, C}, g.options());
    check_rc(lyc_col2im_rows(cptr(dx_rows), mptr(dxr), B, C, H, W, gm.kh, gm.kw, gm.sh, gm.sw, gm.ph, gm.pw, gm.dh, gm.dw,
                             dtype_code(g.scalar_type()) | (f32_rows ? LYC_F32_ROWS : 0), stream_of(g)), "lyc_col2im_rows");
    return from_rows(dxr, B, H, W, x_cl);
  }
  Tensor dx = at::empty(xshape, g.options());
  check_rc(lyc_col2im(cptr(dx_rows), mptr(dx), B, xshape[1], H, W, gm.kh, gm.kw, gm.sh, gm.sw, gm.ph, gm.pw, gm.dh, gm.dw,
                      dtype_code(g.scalar_type()) | (f32_rows ? LYC_F32_ROWS : 0), stream_of(g)), "lyc_col2im");
  return dx;
}

// the functional backward op (what a compiled graph calls): gradients as new tensors in the factors' dtypes
std::tuple<Tensor, Tensor, Tensor, Tensor, Tensor> adapter_conv2d_bwd(const Tensor& g, const Tensor& x, const Tensor& cols, const Tensor& saved,
                                                                      const Tensor& f0, const Tensor& f1, const c10::optional<Tensor>& f2,
                                                                      const c10::optional<Tensor>& f3, int64_t algo, double alpha,
                                                                      at::IntArrayRef kernel, at::IntArrayRef stride, at::IntArrayRef padding,
                                                                      at::IntArrayRef dilation, bool need_dx, bool need_f) {
  Geom gm = geom_of(kernel, stride, padding, dilation, x.size(2), x.size(3));
  const bool pw = pointwise(gm);
  Tensor u2 = f2.has_value() ? *f2 : Tensor(), u3 = f3.has_value() ? *f3 : Tensor();
  const Tensor* f[4] = {&f0, &f1, &u2, &u3};
  const int nf = algo == ALGO_LOHA ? 4 : 2;
  Tensor d[4];
  if (need_f)
    for (int i = 0; i < nf; ++i) d[i] = at::zeros(f[i]->sizes(), f[i]->options().dtype(at::kFloat));
  Tensor dx = adapter_conv2d_bwd_into(g, pw ? x : cols, !pw, x.sizes(), rows_are_free(x), saved, f, algo, alpha, gm, need_dx, d);
  auto out = [&](int i) { return (need_f && i < nf) ? d[i].to(f[i]->scalar_type()) : at::empty({0}, f0.options()); };
  return {dx.defined() ? dx : at::empty({0}, x.options()), out(0), out(1), out(2), out(3)};
}

struct AdapterConv2dFn : public torch::autograd::Function<AdapterConv2dFn> {
  static Tensor forward(AutogradContext* ctx, const Tensor& x, const Tensor& f0, const Tensor& f1, const c10::optional<Tensor>& f2,
                        const c10::optional<Tensor>& f3, int64_t algo, double alpha, std::vector<int64_t> kernel, std::vector<int64_t> stride,
                        std::vector<int64_t> padding, std::vector<int64_t> dilation) {
    at::AutoDispatchBelowADInplaceOrView guard;
    static auto op = c10::Dispatcher::singleton().findSchemaOrThrow("lycoris_amd::_adapter_conv2d_forward", "")
                         .typed<std::tuple<Tensor, Tensor, Tensor>(const Tensor&, const Tensor&, const Tensor&, const c10::optional<Tensor>&,
                                                                   const c10::optional<Tensor>&, int64_t, double, at::IntArrayRef,
                                                                   at::IntArrayRef, at::IntArrayRef, at::IntArrayRef)>();
    auto [y, cols, saved] = op.call(x, f0, f1, f2, f3, algo, alpha, kernel, stride, padding, dilation);
    Tensor u2 = f2.has_value() ? *f2 : Tensor(), u3 = f3.has_value() ? *f3 : Tensor();
    const Tensor* fs[4] = {&f0, &f1, &u2, &u3};
    for (const Tensor* f : fs)
      if (f->defined()) expect(*f, x);
    save_vars(ctx, {x, cols, saved, f0, f1, u2, u3});
    ctx->saved_data["algo"] = algo;
    ctx->saved_data["alpha"] = alpha;
    ctx->saved_data["geom"] = std::vector<int64_t>{kernel[0], kernel[1], stride[0], stride[1], padding[0], padding[1], dilation[0], dilation[1]};
    return y;
  }
  static variable_list backward(AutogradContext* ctx, variable_list grads) {
    auto s = saved_vars(ctx);
    const Tensor &x = s[0], &cols = s[1], &saved = s[2];
    const int64_t algo = ctx->saved_data["algo"].toInt();
    if (algo == ALGO_LOKR || algo == 2 /* ALGO_LOHA */) planes_mark_dirty_after_backward();
    const double alpha = ctx->saved_data["alpha"].toDouble();
    auto gv = ctx->saved_data["geom"].toIntVector();
    const std::vector<int64_t> kernel{gv[0], gv[1]}, stride{gv[2], gv[3]}, padding{gv[4], gv[5]}, dilation{gv[6], gv[7]};
    const int nf = algo == ALGO_LOHA ? 4 : 2;
    const bool nx = ctx->needs_input_grad(0);
    bool need[4] = {false, false, false, false}, any = false;
    for (int i = 0; i < nf; ++i) any = (need[i] = ctx->needs_input_grad(1 + i)) || any;
    const Tensor& g = grads[0];
    variable_list out(11);
    if (eager_cuda(g) && eager_cuda(x)) {
      Geom gm = geom_of(kernel, stride, padding, dilation, x.size(2), x.size(3));
      const bool pw = pointwise(gm);
      const Tensor* f[4] = {&s[3], &s[4], &s[5], &s[6]};
      GradTarget t[4];
      Tensor d[4];
      for (int i = 0; i < nf; ++i) {
        t[i] = grad_target(*f[i], need[i]);
        d[i] = t[i].buf;
      }
      Tensor dx = adapter_conv2d_bwd_into(g, pw ? x : cols, !pw, x.sizes(), rows_are_free(x), saved, f, algo, alpha, gm, nx, d);
      out[0] = dx;
      for (int i = 0; i < nf; ++i) out[1 + i] = finish_grad(*f[i], t[i]);
      return out;
    }
    static auto op = c10::Dispatcher::singleton().findSchemaOrThrow("lycoris_amd::_adapter_conv2d_backward", "")
                         .typed<std::tuple<Tensor, Tensor, Tensor, Tensor, Tensor>(
                             const Tensor&, const Tensor&, const Tensor&, const Tensor&, const Tensor&, const Tensor&, const c10::optional<Tensor>&,
                             const c10::optional<Tensor>&, int64_t, double, at::IntArrayRef, at::IntArrayRef, at::IntArrayRef, at::IntArrayRef,
                             bool, bool)>();
    c10::optional<Tensor> o2 = s[5].defined() ? c10::optional<Tensor>(s[5]) : c10::nullopt;
    c10::optional<Tensor> o3 = s[6].defined() ? c10::optional<Tensor>(s[6]) : c10::nullopt;
    auto [dx, d0, d1, d2, d3] = op.call(g, x, cols, saved, s[3], s[4], o2, o3, algo, alpha, kernel, stride, padding, dilation, nx, any);
    out[0] = nx ? dx : Tensor();
    const Tensor* dd[4] = {&d0, &d1, &d2, &d3};
    for (int i = 0; i < nf; ++i) out[1 + i] = need[i] ? *dd[i] : Tensor();
    return out;
  }
};
Tensor adapter_conv2d_autograd(const Tensor& x, const Tensor& f0, const Tensor& f1, const c10::optional<Tensor>& f2,
                               const c10::optional<Tensor>& f3, int64_t algo, double alpha, at::IntArrayRef kernel, at::IntArrayRef stride,
                               at::IntArrayRef padding, at::IntArrayRef dilation) {
  const GradAtApply ga_;
  return AdapterConv2dFn::apply(amp(x), f0, f1, f2, f3, algo, alpha, kernel.vec(), stride.vec(), padding.vec(), dilation.vec());
}
Tensor adapter_conv2d_cuda(const Tensor& x, const Tensor& f0, const Tensor& f1, const c10::optional<Tensor>& f2, const c10::optional<Tensor>& f3,
                           int64_t algo, double alpha, at::IntArrayRef kernel, at::IntArrayRef stride, at::IntArrayRef padding,
                           at::IntArrayRef dilation) {
  return std::get<0>(adapter_conv2d_fwd(x, f0, f1, f2, f3, algo, alpha, kernel, stride, padding, dilation));
}
std::tuple<Tensor, Tensor, Tensor> adapter_conv2d_fwd_meta(const Tensor& x, const Tensor& f0, const Tensor& f1, const c10::optional<Tensor>& f2,
                                                           const c10::optional<Tensor>& f3, int64_t algo, double alpha, at::IntArrayRef kernel,
                                                           at::IntArrayRef stride, at::IntArrayRef padding, at::IntArrayRef dilation) {
  Geom gm = geom_of(kernel, stride, padding, dilation, x.size(2), x.size(3));
  const int64_t B = x.size(0), C = x.size(1), O = rows_out_features(algo, f0, f1), M = B * gm.Ho * gm.Wo;
  const bool pw = pointwise(gm);
  Tensor y = at::empty({B, O, gm.Ho, gm.Wo},
                       x.options().memory_format((pw || window_major_cols(x, algo, gm, C)) && rows_are_free(x) ? at::MemoryFormat::ChannelsLast
                                                                                                           : at::MemoryFormat::Contiguous));
  Tensor cols = pw ? x.new_empty({0}) : x.new_empty({M, C * gm.kh * gm.kw});
  Tensor saved;
  if (algo == ALGO_LOKR) saved = x.new_empty({0});
  else if (algo == ALGO_LOCON) saved = x.new_empty({M, f0.size(0)}, x.options().dtype(at::kFloat));
  else saved = x.new_empty({lyc_loha_workspace_bytes((int)O, (int)(C * gm.kh * gm.kw), dtype_code(x.scalar_type()))}, x.options().dtype(at::kByte));
  return {y, cols, saved};
}
Tensor adapter_conv2d_meta(const Tensor& x, const Tensor& f0, const Tensor& f1, const c10::optional<Tensor>& f2, const c10::optional<Tensor>& f3,
                           int64_t algo, double alpha, at::IntArrayRef kernel, at::IntArrayRef stride, at::IntArrayRef padding,
                           at::IntArrayRef dilation) {
  return std::get<0>(adapter_conv2d_fwd_meta(x, f0, f1, f2, f3, algo, alpha, kernel, stride, padding, dilation));
}
std::tuple<Tensor, Tensor, Tensor, Tensor, Tensor> adapter_conv2d_bwd_meta(const Tensor& g, const Tensor& x, const Tensor& cols, const Tensor& saved,
                                                                           const Tensor& f0, const Tensor& f1, const c10::optional<Tensor>& f2,
                                                                           const c10::optional<Tensor>& f3, int64_t algo, double alpha,
                                                                           at::IntArrayRef kernel, at::IntArrayRef stride, at::IntArrayRef padding,
                                                                           at::IntArrayRef dilation, bool need_dx, bool need_f) {
  auto e = [&](const Tensor& t, bool n) { return n ? at::empty_like(t) : t.new_empty({0}); };
  auto eo = [&](const c10::optional<Tensor>& t, bool n) { return (n && t.has_value()) ? at::empty_like(*t) : f0.new_empty({0}); };
  Tensor dx = need_dx ? at::empty(x.sizes(), x.options().memory_format(pointwise(geom_of(kernel, stride, padding, dilation, x.size(2), x.size(3))) && rows_are_free(x)
                                                                             ? at::MemoryFormat::ChannelsLast : at::MemoryFormat::Contiguous))
                      : x.new_empty({0});
  return {dx, e(f0, need_f), e(f1, need_f), eo(f2, need_f), eo(f3, need_f)};
}

}  // namespace

TORCH_LIBRARY(lycoris_amd, m) {
  // public ops: what lycoris_amd.ops / the modules call (autograd-aware)
  m.def("lokr_linear(Tensor x, Tensor w1, Tensor w2, float alpha, Tensor? base=None) -> Tensor");
  m.def("lokr_linear_group(Tensor x, Tensor[] factors, float[] alphas, Tensor[] bases) -> Tensor[]");
  m.def("lokr_linear_lr_group(Tensor x, Tensor[] factors, float[] alphas, Tensor[] bases) -> Tensor[]");
  m.def("lokr_adapted_linear(Tensor x, Tensor[] factors, float[] alphas, Tensor[] weights, Tensor?[] biases) -> Tensor[]");
  m.def("lokr_linear_lr(Tensor x, Tensor w1, Tensor w2a, Tensor w2b, float alpha, Tensor? base=None) -> Tensor");
  m.def("lokr_linear_lr2(Tensor x, Tensor w1a, Tensor w1b, Tensor w2a, Tensor w2b, float alpha, Tensor? base=None) -> Tensor");
  m.def("locon_linear(Tensor x, Tensor down, Tensor up, float alpha) -> Tensor");
  m.def("locon_linear_group(Tensor x, Tensor[] factors, float[] alphas) -> Tensor[]");
  m.def("loha_linear(Tensor x, Tensor w1a, Tensor w1b, Tensor w2a, Tensor w2b, float alpha) -> Tensor");
  m.def("chan_affine(Tensor a, Tensor w, Tensor? bias, float s0, float mult, int chan_dim) -> Tensor");
  m.def("lokr_conv2d(Tensor x, Tensor w1, Tensor w2, float alpha, int[2] stride, int[2] padding, int[2] dilation) -> Tensor");
  m.def("locon_conv2d(Tensor x, Tensor down, Tensor up, float alpha, int[2] stride, int[2] padding, int[2] dilation) -> Tensor");
  m.def("lokr_conv2d_lr(Tensor x, Tensor w1, Tensor w2a, Tensor w2b, float alpha, int[2] kernel, int[2] stride, int[2] padding, "
        "int[2] dilation) -> Tensor");
  m.def("adapter_conv2d(Tensor x, Tensor f0, Tensor f1, Tensor? f2, Tensor? f3, int algo, float alpha, int[2] kernel, int[2] stride, "
        "int[2] padding, int[2] dilation) -> Tensor");
  m.def("_adapter_conv2d_forward(Tensor x, Tensor f0, Tensor f1, Tensor? f2, Tensor? f3, int algo, float alpha, int[2] kernel, int[2] stride, "
        "int[2] padding, int[2] dilation) -> (Tensor, Tensor, Tensor)");
  m.def("_adapter_conv2d_backward(Tensor g, Tensor x, Tensor cols, Tensor saved, Tensor f0, Tensor f1, Tensor? f2, Tensor? f3, int algo, "
        "float alpha, int[2] kernel, int[2] stride, int[2] padding, int[2] dilation, bool need_dx, bool need_f) "
        "-> (Tensor, Tensor, Tensor, Tensor, Tensor)");
  // forward / backward kernels as functional ops (what torch.compile traces through)
  m.def("_lokr_linear_backward(Tensor g, Tensor x, Tensor w1, Tensor w2, float alpha, bool need_dx, bool need_dw1, bool need_dw2) "
        "-> (Tensor, Tensor, Tensor)");
  m.def("_locon_linear_forward(Tensor x, Tensor down, Tensor up, float alpha) -> (Tensor, Tensor)");
  m.def("_locon_linear_backward(Tensor g, Tensor x, Tensor down, Tensor up, Tensor t, float alpha, bool need_dx, bool need_dd, "
        "bool need_du) -> (Tensor, Tensor, Tensor)");
  m.def("_loha_linear_forward(Tensor x, Tensor w1a, Tensor w1b, Tensor w2a, Tensor w2b, float alpha) -> (Tensor, Tensor)");
  m.def("_loha_linear_backward(Tensor g, Tensor x, Tensor w1a, Tensor w1b, Tensor w2a, Tensor w2b, Tensor ws, float alpha, "
        "bool need_dx, bool need_f) -> (Tensor, Tensor, Tensor, Tensor, Tensor)");
  m.def("_chan_reduce(Tensor g, Tensor a, Tensor? bias, Tensor w_like, float mult, int chan_dim) -> Tensor");
  m.def("_lokr_conv2d_backward(Tensor g, Tensor x, Tensor w1, Tensor w2, float alpha, int[2] stride, int[2] padding, int[2] dilation, "
        "bool need_dx, bool need_dw1, bool need_dw2) -> (Tensor, Tensor, Tensor)");
  m.def("_locon_conv2d_forward(Tensor x, Tensor down, Tensor up, float alpha, int[2] stride, int[2] padding, int[2] dilation) "
        "-> (Tensor, Tensor)");
  m.def("_locon_conv2d_backward(Tensor g, Tensor x, Tensor down, Tensor up, Tensor t, float alpha, int[2] stride, int[2] padding, "
        "int[2] dilation, bool need_dx, bool need_dd, bool need_du) -> (Tensor, Tensor, Tensor)");
}

TORCH_LIBRARY_IMPL(lycoris_amd, CUDA, m) {
  m.impl("lokr_linear", lokr_linear_fwd);
  m.impl("lokr_linear_group", lokr_linear_group_fwd);
  m.impl("lokr_linear_lr_group", lokr_linear_lr_group_fwd);
  m.impl("lokr_linear_lr", lokr_linear_lr_fwd);
  m.impl("lokr_linear_lr2", lokr_linear_lr2_cuda);
  m.impl("_lokr_linear_backward", lokr_linear_bwd);
  m.impl("locon_linear", locon_linear_cuda);
  m.impl("locon_linear_group", locon_linear_group_fwd);
  m.impl("_locon_linear_forward", locon_linear_fwd);
  m.impl("_locon_linear_backward", locon_linear_bwd);
  m.impl("loha_linear", loha_linear_cuda);
  m.impl("_loha_linear_forward", loha_linear_fwd);
  m.impl("_loha_linear_backward", loha_linear_bwd);
  m.impl("chan_affine", chan_scale);
  m.impl("_chan_reduce", chan_reduce);
  m.impl("lokr_conv2d", lokr_conv2d_eager);    // inference_mode skips the Autograd key
  m.impl("locon_conv2d", locon_conv2d_eager);
  m.impl("_lokr_conv2d_backward", lokr_conv2d_bwd);
  m.impl("_locon_conv2d_forward", locon_conv2d_fwd);
  m.impl("_locon_conv2d_backward", locon_conv2d_bwd);
  m.impl("adapter_conv2d", adapter_conv2d_cuda);
  m.impl("lokr_conv2d_lr", lokr_conv2d_lr_cuda);
  m.impl("_adapter_conv2d_forward", adapter_conv2d_fwd);
  m.impl("_adapter_conv2d_backward", adapter_conv2d_bwd);
}

TORCH_LIBRARY_IMPL(lycoris_amd, Meta, m) {
  m.impl("lokr_linear", lokr_linear_meta);
  m.impl("lokr_linear_group", lokr_linear_group_meta);
  m.impl("lokr_linear_lr_group", lokr_linear_group_meta);
  m.impl("lokr_linear_lr", lokr_linear_lr_meta);
  m.impl("lokr_linear_lr2", lokr_linear_lr2_meta);
  m.impl("_lokr_linear_backward", lokr_linear_bwd_meta);
  m.impl("locon_linear", locon_linear_meta);
  m.impl("locon_linear_group", locon_linear_group_meta);
  m.impl("_locon_linear_forward", locon_linear_fwd_meta);
  m.impl("_locon_linear_backward", locon_linear_bwd_meta);
  m.impl("loha_linear", loha_linear_meta);
  m.impl("_loha_linear_forward", loha_linear_fwd_meta);
  m.impl("_loha_linear_backward", loha_linear_bwd_meta);
  m.impl("chan_affine", chan_affine_meta);
  m.impl("_chan_reduce", chan_reduce_meta);
  m.impl("lokr_conv2d", lokr_conv2d_meta);
  m.impl("locon_conv2d", locon_conv2d_meta);
  m.impl("_lokr_conv2d_backward", lokr_conv2d_bwd_meta);
  m.impl("_locon_conv2d_forward", locon_conv2d_fwd_meta);
  m.impl("_locon_conv2d_backward", locon_conv2d_bwd_meta);
  m.impl("adapter_conv2d", adapter_conv2d_meta);
  m.impl("lokr_conv2d_lr", lokr_conv2d_lr_meta);
  m.impl("_adapter_conv2d_forward", adapter_conv2d_fwd_meta);
  m.impl("_adapter_conv2d_backward", adapter_conv2d_bwd_meta);
}

TORCH_LIBRARY_IMPL(lycoris_amd, Autograd, m) {
  m.impl("lokr_linear", lokr_linear_autograd);
  m.impl("lokr_linear_group", lokr_linear_group_autograd);
  m.impl("lokr_linear_lr_group", lokr_linear_lr_group_autograd);
  m.impl("lokr_adapted_linear", lokr_adapted_linear_autograd);
  m.impl("lokr_linear_lr", lokr_linear_lr_autograd);
  m.impl("lokr_linear_lr2", lokr_linear_lr2_autograd);
  m.impl("locon_linear", locon_linear_autograd);
  m.impl("locon_linear_group", locon_linear_group_autograd);
  m.impl("loha_linear", loha_linear_autograd);
  m.impl("chan_affine", chan_affine_autograd);
  m.impl("lokr_conv2d", lokr_conv2d_implicit);
  m.impl("locon_conv2d", locon_conv2d_implicit);
  m.impl("adapter_conv2d", adapter_conv2d_autograd);
  m.impl("lokr_conv2d_lr", lokr_conv2d_lr_autograd);
}

void lyc_bind_rccl(py::module_& m);  // rccl_comm.cpp: the ProcessGroup-free communicator of the DP gradient exchange

PYBIND11_MODULE(_lyc_torch, m) {
  m.doc() = "lycoris_amd: TORCH_LIBRARY(lycoris_amd) custom ops over liblycoris_amd.so";
  lyc_bind_rccl(m);
  m.def("set_accum", [](bool enabled, py::object callback, py::object batch_callback) {
    std::lock_guard<std::mutex> lk(g_accum.mu);
    g_accum.enabled = enabled;
    if (g_accum.callback == nullptr) g_accum.callback = new py::object();
    if (g_accum.batch_callback == nullptr) g_accum.batch_callback = new py::object();
    g_accum.has_callback = !callback.is_none();
    g_accum.has_batch = !batch_callback.is_none();
    *g_accum.callback = std::move(callback);
    *g_accum.batch_callback = std::move(batch_callback);
    g_accum.uses.clear();
  });
  m.def("lokr_conv2d_lr_ok", [](const Tensor& x, const Tensor& w1, const Tensor& w2a, const Tensor& w2b, std::vector<int64_t> kernel,
                                std::vector<int64_t> stride, std::vector<int64_t> padding, std::vector<int64_t> dilation) {
    return lokr_conv2d_lr_ok(x, w1, w2a, w2b, kernel, stride, padding, dilation);
  });
  m.def("set_planes_cache", [](bool enabled) {
    g_planes.enabled = enabled;
    if (!enabled) {
      std::lock_guard<std::mutex> lk(g_planes.mu);
      g_planes.map.clear();
    }
  });
  m.def("planes_cache_size", []() {
    std::lock_guard<std::mutex> lk(g_planes.mu);
    return g_planes.map.size();
  });
  m.def("refresh_planes", [](bool force) {  // every cached plane set whose parameter changed (force: all), on the current streams
    std::lock_guard<std::mutex> lk(g_planes.mu);
    // an explicit refresh in front of a forward pass IS the step boundary: take the pending "dirty" mark here, so that the first layer
    // call behind it does not find a new epoch and refresh everything a second time (round 6: every replay of a captured step did)
    planes_new_epoch_if_dirty_locked(true);
    std::vector<c10::DeviceIndex> devs;
    for (auto& kv : g_planes.map)
      if (std::find(devs.begin(), devs.end(), kv.second.device) == devs.end()) devs.push_back(kv.second.device);
    for (auto& kv : g_loha_planes)
      if (std::find(devs.begin(), devs.end(), kv.second.device) == devs.end()) devs.push_back(kv.second.device);
    for (c10::DeviceIndex dv : devs) {
      refresh_planes_locked(dv, c10::hip::getCurrentHIPStream(dv).stream(), force);
      refresh_loha_planes_locked(dv, c10::hip::getCurrentHIPStream(dv).stream(), force);
    }
  }, py::arg("force") = false);
  m.def("reset_use_counts", []() {  // once per optimizer step (AdapterGradSync.zero_grad / finish): drop counts of forwards
    std::lock_guard<std::mutex> lk(g_accum.mu);  // whose backward never ran
    g_accum.uses.clear();
    g_accum.done_task.clear();
  });
  m.def("debug_recompute_state", []() { return std::make_tuple(torch::autograd::get_current_graph_task_id(), at::SavedTensorDefaultHooks::is_enabled(), (int)at::SavedTensorDefaultHooks::get_tls_state().stack.size(), inside_checkpoint_recompute()); });
  m.def("fused_reports", [](const Tensor& p) { return fused_reports(p); });
  m.def("mark_planes_dirty", []() { g_planes.dirty = true; });  // step boundary (optimizer-step post hook, ops.py)
  m.def("planes_epoch", []() {
    std::lock_guard<std::mutex> lk(g_planes.mu);
    return g_planes.epoch;
  });
  m.def("accum_enabled", []() { return g_accum.enabled; });
  m.def("set_defer", [](bool enabled, int flush_at) {
    g_defer.enabled = enabled;
    if (flush_at > 0) g_defer.flush_at = (size_t)flush_at;
  }, py::arg("enabled"), py::arg("flush_at") = 0);
  m.def("defer_enabled", []() { return g_defer.enabled.load(); });
  m.def("deferred_pending", []() {
    size_t n = 0;
    for (DeferredLists& L : g_defer.dev) {
      std::lock_guard<std::mutex> lk(L.mu);
      n += L.size();
    }
    return n;
  });
  m.def("flush_deferred", []() {
    py::gil_scoped_release nogil;  // flush_deferred() notifies through a callback that takes the GIL itself
    for (int d = 0; d < kMaxDevices; ++d) flush_deferred((c10::DeviceIndex)d);
  });
  m.def("discard_deferred", []() {  // after a failed backward: drop parked layers instead of adding them to the next step
    size_t n = 0;
    for (DeferredLists& L : g_defer.dev) {
      std::vector<DeferredLokr> items;
      std::vector<DeferredLocon> litems;
      std::vector<DeferredLoha> hitems;
      std::vector<DeferredLokrConv> citems;
      {
        std::lock_guard<std::mutex> lk(L.mu);
        items.swap(L.lokr);
        litems.swap(L.locon);
        hitems.swap(L.loha);
        citems.swap(L.lokr_conv);
        L.callback_queued = false;
      }
      n += items.size() + litems.size() + hitems.size() + citems.size();  // the tensors are released outside the lock
    }
    return n;
  });
  m.def("abi_version", []() { return lyc_abi_version(); });
  m.def("locon_reg_staged", [](bool on) { const bool was = g_locon_reg; g_locon_reg = on; return was; });
}
