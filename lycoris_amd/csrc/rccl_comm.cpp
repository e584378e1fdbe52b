// rccl_comm.cpp -- a ProcessGroup-free RCCL communicator for the adapter-gradient exchange ("DP path: a separate
// ProcessGroup-free RCCL communicator object created once per process").
//
// The reference has no code for this step: under sd-scripts the network is wrapped in DistributedDataParallel (accelerate,
// /root/reference/requirements-kohya.txt:43).  Round 3-4 ran the bucket collectives through c10d's ProcessGroupNCCL; that costs a
// watchdog thread, a Work object + two events per collective and (measured at world_size 1, profiles/r04_final_ws1.log) +1.9 ms on a
// 19.6 ms step for 6 collectives whose GPU time is 0.4 ms.  Here a collective is nothing but stream work:
//
//     ncclCommInitRank once (the 128-byte id travels through whatever key-value store the launcher provides)
//     one HIP stream owned by the communicator (default priority: a HIGH-priority stream made every kernel of the step slower on
//     this stack -- 73.7 against 21.0 ms per SDXL step, profiles/r05_ws1_stream_and_event_ab.log)
//     wait_current() / wait_event(e): the communicator's stream waits for the gradients of the bucket
//     all_reduce / reduce_scatter / all_gather: ncclXxx(..., comm, stream) -- in place on the gradient arena
//     join(): the caller's stream waits for everything enqueued so far
//
// No host thread, no Work objects, no host synchronisation.  Because it is plain stream work it may also be recorded into a
// hipGraph: when the caller's stream is capturing, wait_current() forks the communicator's stream into the capture and join() joins
// it back (bench.py --capture-collectives).
//
// Part of _lyc_torch.so (host-only C++; the kernels are RCCL's).  One process per GPU: a communicator is bound to one device.
#include <ATen/ATen.h>
#include <c10/hip/HIPStream.h>
#include <torch/extension.h>

#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

#include <cstring>
#include <string>

namespace {

using at::Tensor;

#define LYC_NCCL(call)                                                                                         \
  do {                                                                                                         \
    ncclResult_t r_ = (call);                                                                                  \
    TORCH_CHECK(r_ == ncclSuccess, "lycoris_amd RCCL: " #call " failed: ", ncclGetErrorString(r_));         \
  } while (0)
#define LYC_HIP(call)                                                                                          \
  do {                                                                                                         \
    hipError_t e_ = (call);                                                                                    \
    TORCH_CHECK(e_ == hipSuccess, "lycoris_amd RCCL: " #call " failed: ", hipGetErrorString(e_));           \
  } while (0)

ncclDataType_t nccl_type(const Tensor& t) {
  switch (t.scalar_type()) {
    case at::kFloat: return ncclFloat32;
    case at::kDouble: return ncclFloat64;
    case at::kHalf: return ncclFloat16;
    case at::kBFloat16: return ncclBfloat16;
    case at::kInt: return ncclInt32;
    case at::kLong: return ncclInt64;
    default: TORCH_CHECK(false, "lycoris_amd RCCL: unsupported dtype ", t.scalar_type());
  }
}
ncclRedOp_t nccl_op(int op) {
  switch (op) {
    case 0: return ncclSum;
    case 1: return ncclAvg;
    case 2: return ncclMax;
    default: TORCH_CHECK(false, "lycoris_amd RCCL: op must be 0 (sum), 1 (avg) or 2 (max)");
  }
}

class RcclComm {
 public:
  RcclComm(const std::string& id, int rank, int world, int device, bool high_priority, uintptr_t external_stream, bool on_current_stream)
      : rank_(rank), world_(world), device_(device), on_current_(on_current_stream) {
    TORCH_CHECK(id.size() == sizeof(ncclUniqueId), "lycoris_amd RCCL: the unique id must be ", sizeof(ncclUniqueId), " bytes");
    TORCH_CHECK(world >= 1 && rank >= 0 && rank < world, "lycoris_amd RCCL: rank ", rank, " of ", world);
    ncclUniqueId uid;
    std::memcpy(&uid, id.data(), sizeof uid);
    int prev = 0;
    LYC_HIP(hipGetDevice(&prev));
    LYC_HIP(hipSetDevice(device));
    try {
      int lo = 0, hi = 0;
      LYC_HIP(hipDeviceGetStreamPriorityRange(&lo, &hi));  // hi = numerically lowest = highest priority
      if (external_stream != 0) {  // a stream the caller owns (e.g. one of torch's pool streams) and keeps alive
        stream_ = reinterpret_cast<hipStream_t>(external_stream);
        own_stream_ = false;
      } else {
        LYC_HIP(hipStreamCreateWithPriority(&stream_, hipStreamNonBlocking, high_priority ? hi : 0));
      }
      // Ordering events with a DEVICE-scope release: a default HIP event performs a system-scope release when it is recorded (L2
      // write-back for host visibility) -- measured on the SDXL step: five default-flag events between the backward segments cost
      // +1.5 ms per step with collectives that do nothing at all (profiles/r05_ws1_stream_and_event_ab.log).  All consumers of these
      // events are streams of the same device; what crosses devices is RCCL's own business.
      constexpr unsigned kFlags = hipEventDisableTiming | hipEventReleaseToDevice;
      LYC_HIP(hipEventCreateWithFlags(&ev_in_, kFlags));
      LYC_HIP(hipEventCreateWithFlags(&ev_out_, kFlags));
      for (hipEvent_t& e : marks_) LYC_HIP(hipEventCreateWithFlags(&e, kFlags));
      {
        py::gil_scoped_release nogil;  // the bootstrap blocks until every rank has arrived
        LYC_NCCL(ncclCommInitRank(&comm_, world, uid, rank));
      }
    } catch (...) {  // a failed bootstrap leaves nothing behind: stream, events, and the caller's device
      release_handles();
      (void)hipSetDevice(prev);
      throw;
    }
    LYC_HIP(hipSetDevice(prev));
  }
  ~RcclComm() { destroy(); }

  void destroy() {
    if (comm_ != nullptr) {
      (void)hipStreamSynchronize(stream_);
      (void)ncclCommDestroy(comm_);
      comm_ = nullptr;
      release_handles();
    }
  }
  // the number of ranks RCCL itself reports for this communicator (ncclCommCount): bench.py prints it as `rccl_ranks`
  int count() const {
    TORCH_CHECK(comm_ != nullptr, "lycoris_amd RCCL: communicator destroyed");
    int n = 0;
    LYC_NCCL(ncclCommCount(comm_, &n));
    return n;
  }

  int rank() const { return rank_; }
  int world() const { return world_; }
  int device() const { return device_; }
  uintptr_t stream() const { return reinterpret_cast<uintptr_t>(stream_); }
  bool on_current_stream() const { return on_current_; }
  // the stream a collective is enqueued on: the communicator's own, or (on_current_stream) whatever stream is current at the call --
  // the exchange is then ordinary in-order work of the compute stream: no events, no second queue (see grad_sync.py for when that wins)
  hipStream_t launch_stream() const { return on_current_ ? c10::hip::getCurrentHIPStream(device_).stream() : stream_; }

  // the communicator's stream waits for everything enqueued so far on the caller's current stream
  void wait_current() {
    if (on_current_) return;
    hipStream_t cur = c10::hip::getCurrentHIPStream(device_).stream();
    LYC_HIP(hipEventRecord(ev_in_, cur));
    LYC_HIP(hipStreamWaitEvent(stream_, ev_in_, 0));
  }
  // mark(): remember "everything enqueued so far on the caller's current stream" (one of a ring of kMarks events; returns its slot),
  // wait_mark(slot): the communicator's stream waits for exactly that point -- later work of the caller's stream is NOT waited for
  int mark() {
    hipStream_t cur = c10::hip::getCurrentHIPStream(device_).stream();
    const int slot = next_mark_;
    next_mark_ = (next_mark_ + 1) % kMarks;
    LYC_HIP(hipEventRecord(marks_[slot], cur));
    return slot;
  }
  void wait_mark(int slot) {
    if (on_current_) return;
    TORCH_CHECK(slot >= 0 && slot < kMarks, "lycoris_amd RCCL: bad mark ", slot);
    LYC_HIP(hipStreamWaitEvent(stream_, marks_[slot], 0));
  }
  // ... or for an event the caller recorded earlier (torch.cuda.Event.cuda_event): later work of the caller's stream is NOT waited for
  void wait_event(uintptr_t ev) { if (on_current_) return; LYC_HIP(hipStreamWaitEvent(stream_, reinterpret_cast<hipEvent_t>(ev), 0)); }
  // the caller's current stream waits for everything enqueued so far on the communicator's stream
  void join() {
    if (on_current_) return;
    hipStream_t cur = c10::hip::getCurrentHIPStream(device_).stream();
    LYC_HIP(hipEventRecord(ev_out_, stream_));
    LYC_HIP(hipStreamWaitEvent(cur, ev_out_, 0));
  }
  void synchronize() {
    py::gil_scoped_release nogil;
    LYC_HIP(hipStreamSynchronize(launch_stream()));
  }

  void check(const Tensor& t, const char* what) const {
    TORCH_CHECK(t.is_cuda() && t.device().index() == device_, "lycoris_amd RCCL: ", what, " must live on cuda:", device_);
    TORCH_CHECK(t.is_contiguous(), "lycoris_amd RCCL: ", what, " must be contiguous");
  }
  // in place
  void all_reduce(const Tensor& t, int op) {
    check(t, "all_reduce buffer");
    LYC_NCCL(ncclAllReduce(t.const_data_ptr(), t.mutable_data_ptr(), (size_t)t.numel(), nccl_type(t), nccl_op(op), comm_, launch_stream()));
  }
  // shard = reduce over the ranks of full[rank * n, (rank + 1) * n); `shard` may be that very slice of `full` (in place)
  void reduce_scatter(const Tensor& shard, const Tensor& full, int op) {
    check(shard, "reduce_scatter shard");
    check(full, "reduce_scatter buffer");
    TORCH_CHECK(full.numel() == shard.numel() * world_ && full.scalar_type() == shard.scalar_type(),
                "lycoris_amd RCCL: reduce_scatter needs full.numel() == world * shard.numel()");
    LYC_NCCL(ncclReduceScatter(full.const_data_ptr(), shard.mutable_data_ptr(), (size_t)shard.numel(), nccl_type(full), nccl_op(op), comm_, launch_stream()));
  }
  void all_gather(const Tensor& full, const Tensor& shard) {
    check(shard, "all_gather shard");
    check(full, "all_gather buffer");
    TORCH_CHECK(full.numel() == shard.numel() * world_ && full.scalar_type() == shard.scalar_type(),
                "lycoris_amd RCCL: all_gather needs full.numel() == world * shard.numel()");
    LYC_NCCL(ncclAllGather(shard.const_data_ptr(), full.mutable_data_ptr(), (size_t)shard.numel(), nccl_type(full), comm_, launch_stream()));
  }
  void broadcast(const Tensor& t, int root) {
    check(t, "broadcast buffer");
    LYC_NCCL(ncclBroadcast(t.const_data_ptr(), t.mutable_data_ptr(), (size_t)t.numel(), nccl_type(t), root, comm_, launch_stream()));
  }
  void group_start() { LYC_NCCL(ncclGroupStart()); }
  void group_end() { LYC_NCCL(ncclGroupEnd()); }

 private:
  void release_handles() {
    if (ev_in_ != nullptr) (void)hipEventDestroy(ev_in_);
    if (ev_out_ != nullptr) (void)hipEventDestroy(ev_out_);
    for (hipEvent_t& e : marks_) {
      if (e != nullptr) (void)hipEventDestroy(e);
      e = nullptr;
    }
    if (own_stream_ && stream_ != nullptr) (void)hipStreamDestroy(stream_);
    ev_in_ = ev_out_ = nullptr;
    stream_ = nullptr;
  }
  ncclComm_t comm_ = nullptr;
  hipStream_t stream_ = nullptr;
  hipEvent_t ev_in_ = nullptr, ev_out_ = nullptr;
  bool own_stream_ = true;
  bool on_current_ = false;
  static constexpr int kMarks = 64;
  hipEvent_t marks_[kMarks] = {};
  int next_mark_ = 0;
  int rank_, world_, device_;
};

}  // namespace

void lyc_bind_rccl(py::module_& m) {
  m.def("rccl_unique_id", []() {
    ncclUniqueId uid;
    LYC_NCCL(ncclGetUniqueId(&uid));
    return py::bytes(reinterpret_cast<const char*>(&uid), sizeof uid);
  });
  m.def("rccl_version", []() {
    int v = 0;
    LYC_NCCL(ncclGetVersion(&v));
    return v;
  });
  py::class_<RcclComm>(m, "RcclComm")
      .def(py::init([](py::bytes id, int rank, int world, int device, bool high_priority, uintptr_t external_stream, bool on_current_stream) {
             return new RcclComm(std::string(id), rank, world, device, high_priority, external_stream, on_current_stream);
           }),
           py::arg("unique_id"), py::arg("rank"), py::arg("world"), py::arg("device"), py::arg("high_priority") = false,
           py::arg("external_stream") = 0, py::arg("on_current_stream") = false)
      .def_property_readonly("on_current_stream", &RcclComm::on_current_stream)
      .def_property_readonly("rank", &RcclComm::rank)
      .def_property_readonly("world", &RcclComm::world)
      .def_property_readonly("device", &RcclComm::device)
      .def_property_readonly("stream", &RcclComm::stream)
      .def("wait_current", &RcclComm::wait_current)
      .def("wait_event", &RcclComm::wait_event)
      .def("mark", &RcclComm::mark)
      .def("wait_mark", &RcclComm::wait_mark)
      .def("join", &RcclComm::join)
      .def("synchronize", &RcclComm::synchronize)
      .def("all_reduce", &RcclComm::all_reduce, py::arg("tensor"), py::arg("op") = 0)
      .def("reduce_scatter", &RcclComm::reduce_scatter, py::arg("shard"), py::arg("full"), py::arg("op") = 0)
      .def("all_gather", &RcclComm::all_gather, py::arg("full"), py::arg("shard"))
      .def("broadcast", &RcclComm::broadcast, py::arg("tensor"), py::arg("root") = 0)
      .def("group_start", &RcclComm::group_start)
      .def("group_end", &RcclComm::group_end)
      .def("count", &RcclComm::count)
      .def("destroy", &RcclComm::destroy);
}
