// sharded_host.hpp -- C++ host of the segment-sharded solve: the C ABI's phase functions + RCCL called directly
// (one all-gather of the interface records per iteration over xGMI; with landmarks one
// all-reduce of the landmark Schur complement).  Header-only; needs <rccl/rccl.h> and the HIP runtime API, i.e. it is
// compiled by the application that owns the communicator, not into libgpslam_hip.so (whose ABI stays free of RCCL types).
//
//   ShardedRank    one rank: a handle created with {rank, nranks}, a communicator and the stream the handle runs on.
//                  enqueue_*() put kernels and collectives on that stream in order; nothing synchronises the host
//                  except stats().  This is what each process of a one-process-per-GPU job owns.
//   ShardedDriver  one process driving all local devices (ncclCommInitAll): the phases of all ranks are issued
//                  back to back and the collectives inside ncclGroupStart / ncclGroupEnd.  Used by the C++ test on
//                  whatever number of GPUs the box has (1 on the build farm), and by single-process deployments.
// The Python twin is gpslam_amd/sharded.py (torch.distributed); both follow gpslam_hip_iterate_phase1/2a/2b.
#pragma once

#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <deque>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/gpslam_hip.h"

namespace gpslam_hip {

inline void hip_ok(hipError_t e, const char *what) {
  if (e != hipSuccess) throw std::runtime_error(std::string(what) + ": " + hipGetErrorString(e));
}
inline void nccl_ok(ncclResult_t r, const char *what) {
  if (r != ncclSuccess) throw std::runtime_error(std::string(what) + ": " + ncclGetErrorString(r));
}

class ShardedRank {
 public:
  /// h: compiled handle of this rank's segment (cfg.rank / cfg.nranks set, halo state set); stream: the stream RCCL uses
  ShardedRank(gpslam_hip_handle *h, int rank, int nranks, int device, ncclComm_t comm, hipStream_t stream)
      : h_(h), rank_(rank), nranks_(nranks), device_(device), comm_(comm), stream_(stream) {
    ok(gpslam_hip_set_stream(h_, (void *)stream_), "set_stream");      // kernels and collectives share one stream: ordered
    ok(gpslam_hip_interface_send(h_, &send_, &send_bytes_), "interface_send");
    ok(gpslam_hip_interface_recv(h_, &recv_, &recv_bytes_), "interface_recv");
    ok(gpslam_hip_landmark_reduce_buffer(h_, &lm_, &lm_bytes_), "landmark_reduce_buffer");
  }
  bool has_landmarks() const { return lm_bytes_ > 0; }
  void phase1(double lambda) { use(); ok(gpslam_hip_iterate_phase1(h_, lambda), "iterate_phase1"); }
  /// the ONE data-path collective of a Gauss-Newton iteration: 3.6 KB per rank for Pose3
  void all_gather() { use(); nccl_ok(ncclAllGather(send_, recv_, send_bytes_, ncclChar, comm_, stream_), "ncclAllGather"); }
  void phase2a() { use(); ok(gpslam_hip_iterate_phase2a(h_), "iterate_phase2a"); }
  void all_reduce_landmarks() {
    use();
    if (lm_bytes_) nccl_ok(ncclAllReduce(lm_, lm_, lm_bytes_ / sizeof(double), ncclDouble, ncclSum, comm_, stream_), "ncclAllReduce");
  }
  /// st == nullptr: no error pass and no host synchronisation (the benchmark loop)
  void phase2b(gpslam_hip_stats *st) { use(); ok(gpslam_hip_iterate_phase2b(h_, st), "iterate_phase2b"); }
  /// Levenberg-Marquardt across the ranks (include/gpslam_hip.h, gpslam_hip_lm_begin ...): the pieces of a trial
  void lm_begin() { use(); ok(gpslam_hip_lm_begin(h_), "lm_begin"); }
  void lm_trial_phase1(double lambda) { use(); ok(gpslam_hip_lm_trial_phase1(h_, lambda), "lm_trial_phase1"); }
  void lm_trial_phase2(double *out6) { use(); ok(gpslam_hip_lm_trial_phase2(h_, out6), "lm_trial_phase2"); }
  void lm_reject() { use(); ok(gpslam_hip_lm_reject(h_), "lm_reject"); }
  /// One process PER rank (the deployment shape: a launcher starts one process per GPU): hand this rank's RCCL communicator to
  /// the library (gpslam_hip_set_collectives).  From then on the optimiser loops of the C ABI -- gpslam_hip_iterate_gn,
  /// gpslam_hip_iterate_lm, gpslam_hip_optimize, gpslam_hip_error -- run on handle() and return the whole chain's statistics, the
  /// collectives being ncclAllGather / ncclAllReduce on this rank's stream.  (Not for ShardedDriver, where ONE thread drives
  /// every rank: a rank's call would wait in the collective for ranks that same thread has not started yet.)  The object must
  /// outlive the handle's use: the library keeps `this` as the callbacks' user pointer.
  void register_collectives() {
    ok(gpslam_hip_set_collectives(h_, &ShardedRank::gather_cb, &ShardedRank::reduce_cb, this), "set_collectives");
  }
  gpslam_hip_handle *handle() const { return h_; }
  int device() const { return device_; }
  hipStream_t stream() const { return stream_; }

 private:
  static int gather_cb(void *user, const void *send, void *recv, size_t bytes_per_rank, void *stream) {
    const ShardedRank *self = static_cast<const ShardedRank *>(user);
    return ncclAllGather(send, recv, bytes_per_rank, ncclChar, self->comm_, (hipStream_t)stream) == ncclSuccess ? 0 : 1;
  }
  static int reduce_cb(void *user, void *buf, size_t n_doubles, void *stream) {
    const ShardedRank *self = static_cast<const ShardedRank *>(user);
    return ncclAllReduce(buf, buf, n_doubles, ncclDouble, ncclSum, self->comm_, (hipStream_t)stream) == ncclSuccess ? 0 : 1;
  }
  void use() const { hip_ok(hipSetDevice(device_), "hipSetDevice"); }
  void ok(int rc, const char *what) const {
    if (rc < 0) throw std::runtime_error(std::string(what) + " failed on rank " + std::to_string(rank_) + ": " + gpslam_hip_last_error(h_));
  }
  gpslam_hip_handle *h_;
  int rank_, nranks_, device_;
  ncclComm_t comm_;
  hipStream_t stream_;
  void *send_ = nullptr, *recv_ = nullptr, *lm_ = nullptr;
  size_t send_bytes_ = 0, recv_bytes_ = 0, lm_bytes_ = 0;
};

/// One process, all local devices: rank r lives on device devices[r].
class ShardedDriver {
 public:
  explicit ShardedDriver(const std::vector<int> &devices) : devices_(devices), comms_(devices.size()), streams_(devices.size()) {
    nccl_ok(ncclCommInitAll(comms_.data(), (int)devices_.size(), devices_.data()), "ncclCommInitAll");
    for (size_t r = 0; r < devices_.size(); r++) {
      hip_ok(hipSetDevice(devices_[r]), "hipSetDevice");
      hip_ok(hipStreamCreateWithFlags(&streams_[r], hipStreamNonBlocking), "hipStreamCreate");
    }
  }
  ~ShardedDriver() {
    for (size_t r = 0; r < devices_.size(); r++) {
      (void)hipSetDevice(devices_[r]);
      (void)hipStreamSynchronize(streams_[r]);
      (void)ncclCommDestroy(comms_[r]);
      (void)hipStreamDestroy(streams_[r]);
    }
  }
  int nranks() const { return (int)devices_.size(); }
  ShardedRank &rank(int r) { return ranks_.at((size_t)r); }
  /// register rank r's compiled handle (created on devices[r] with {rank = r, nranks})
  void add(gpslam_hip_handle *h) {
    const int r = (int)ranks_.size();
    if (r >= nranks()) throw std::invalid_argument("more handles than ranks");
    ranks_.emplace_back(h, r, nranks(), devices_[r], comms_[r], streams_[r]);
  }
  /// one Gauss-Newton (lambda = 0) / damped iteration over all ranks; returns the global statistics when asked
  gpslam_hip_stats iterate(double lambda = 0.0, bool want_stats = true) {
    if ((int)ranks_.size() != nranks()) throw std::invalid_argument("register one handle per rank first");
    for (ShardedRank &r : ranks_) r.phase1(lambda);
    nccl_ok(ncclGroupStart(), "ncclGroupStart");
    for (ShardedRank &r : ranks_) r.all_gather();
    nccl_ok(ncclGroupEnd(), "ncclGroupEnd");
    for (ShardedRank &r : ranks_) r.phase2a();
    if (ranks_[0].has_landmarks() && nranks() > 1) {
      nccl_ok(ncclGroupStart(), "ncclGroupStart");
      for (ShardedRank &r : ranks_) r.all_reduce_landmarks();
      nccl_ok(ncclGroupEnd(), "ncclGroupEnd");
    }
    gpslam_hip_stats tot;
    std::fill((char *)&tot, (char *)&tot + sizeof(tot), 0);
    for (ShardedRank &r : ranks_) {
      gpslam_hip_stats st;
      r.phase2b(want_stats ? &st : nullptr);
      if (want_stats) {      // sums / maxima over the ranks: what the all-gather of the scalars does across processes
        tot.error_before += st.error_before;
        tot.error_after += st.error_after;
        tot.delta_inf_norm = std::max(tot.delta_inf_norm, st.delta_inf_norm);
        tot.status = std::min(tot.status, st.status);
      }
    }
    tot.iterations = 1;
    tot.accepted = 1;
    return tot;
  }
  void synchronize() {
    for (size_t r = 0; r < devices_.size(); r++) {
      hip_ok(hipSetDevice(devices_[r]), "hipSetDevice");
      hip_ok(hipStreamSynchronize(streams_[r]), "hipStreamSynchronize");
    }
  }
  /// LevenbergMarquardtOptimizer::iterate over all ranks (round 4; gpslam_hip_iterate_lm refuses sharded handles because the
  /// accept / reject decision needs global sums): linearise once, then per trial ONE data-path collective -- the all-gather of
  /// the interface records -- (+ the landmark all-reduce where the chain has a dense border), and the six scalars of every
  /// rank {error, trial error, |delta|_inf, delta . g, |delta|^2, indefinite flag} summed / maximised here on the host, in
  /// rank order (one process drives all ranks; across processes they travel in one all-gather of 48 bytes per rank:
  /// gpslam_amd/sharded.py).  Same decisions, same lambda schedule as gpslam_hip_iterate_lm on the unsharded chain.
  gpslam_hip_stats iterate_lm(double *lambda, const gpslam_hip_params &p) {
    if ((int)ranks_.size() != nranks()) throw std::invalid_argument("register one handle per rank first");
    for (ShardedRank &r : ranks_) r.lm_begin();
    gpslam_hip_stats st;
    std::fill((char *)&st, (char *)&st + sizeof(st), 0);
    bool accepted = false;
    int trials = 0;
    double err0 = 0.0, new_err = 0.0, dinf = 0.0, last_trial = 0.0;
    for (;;) {
      for (ShardedRank &r : ranks_) r.lm_trial_phase1(*lambda);
      nccl_ok(ncclGroupStart(), "ncclGroupStart");
      for (ShardedRank &r : ranks_) r.all_gather();
      nccl_ok(ncclGroupEnd(), "ncclGroupEnd");
      for (ShardedRank &r : ranks_) r.phase2a();
      if (ranks_[0].has_landmarks() && nranks() > 1) {
        nccl_ok(ncclGroupStart(), "ncclGroupStart");
        for (ShardedRank &r : ranks_) r.all_reduce_landmarks();
        nccl_ok(ncclGroupEnd(), "ncclGroupEnd");
      }
      double s[6] = {0, 0, 0, 0, 0, 0};
      for (ShardedRank &r : ranks_) {
        double o[6];
        r.lm_trial_phase2(o);
        s[0] += o[0]; s[1] += o[1]; s[3] += o[3]; s[4] += o[4];
        s[2] = std::max(s[2], o[2]); s[5] = std::max(s[5], o[5]);
      }
      err0 = s[0];
      last_trial = s[5] == 0.0 ? s[1] : s[0];
      trials++;
      int32_t keep = 0, done = 0;
      if (gpslam_hip_lm_decide(s, &p, lambda, &keep, &done) < 0) throw std::invalid_argument("gpslam_hip_lm_decide");   // the branch gpslam_hip_iterate_lm takes
      if (keep) { accepted = true; new_err = s[1]; dinf = s[2]; break; }
      for (ShardedRank &r : ranks_) r.lm_reject();
      if (done) break;          // small cost change (lambda untouched) or lambda at its upper bound
    }
    st.error_before = err0;
    st.error_after = accepted ? new_err : err0;
    st.delta_inf_norm = accepted ? dinf : 0.0;
    st.lambda = *lambda;
    st.iterations = 1;
    st.accepted = accepted ? 1 : 0;
    st.trials = trials;
    st.last_trial_error = last_trial;
    return st;
  }

 private:
  std::vector<int> devices_;
  std::vector<ncclComm_t> comms_;
  std::vector<hipStream_t> streams_;
  std::deque<ShardedRank> ranks_;     // (a deque: add() never moves a rank -- register_collectives() hands the library a rank's address)
};

// ---- chains with many locally visible landmarks (BASELINE config 4) across GPUs: pieces joined at shared cut states
// (include/gpslam_hip.h, gpslam_hip_fs_set_split; the Python twin is gpslam_amd/sharded.py: SplitSolver).
//   SplitDriver   one process over the pieces' devices.  Distinct devices: ncclCommInitAll + one grouped ncclAllGather of the
//                 interface records per iteration.  All pieces on ONE device (the build farm, single-GPU runs of a chain
//                 that was cut for other reasons): the gather is P * P device copies on one stream, no communicator.
class SplitDriver {
 public:
  explicit SplitDriver(const std::vector<int> &devices) : devices_(devices), streams_(devices.size(), nullptr) {
    local_ = true;
    for (int d : devices_) local_ = local_ && d == devices_[0];
    if (!local_) {
      comms_.resize(devices_.size());
      nccl_ok(ncclCommInitAll(comms_.data(), (int)devices_.size(), devices_.data()), "ncclCommInitAll");
    }
    for (size_t r = 0; r < devices_.size(); r++) {
      hip_ok(hipSetDevice(devices_[r]), "hipSetDevice");
      if (local_ && r > 0) { streams_[r] = streams_[0]; continue; }     // one device: one stream orders everything
      hip_ok(hipStreamCreateWithFlags(&streams_[r], hipStreamNonBlocking), "hipStreamCreate");
    }
  }
  ~SplitDriver() {
    for (size_t r = 0; r < devices_.size(); r++) {
      (void)hipSetDevice(devices_[r]);
      (void)hipStreamSynchronize(streams_[r]);
      if (!local_) (void)ncclCommDestroy(comms_[r]);
      if (!local_ || r == 0) (void)hipStreamDestroy(streams_[r]);
    }
  }
  int nranks() const { return (int)devices_.size(); }
  /// register piece r's compiled handle (created with nranks = 1 on devices[r], gpslam_hip_fs_set_split(h, r, P, ...) before
  /// compile()); after the last one the pieces agree on the block size of the interface record
  void add(gpslam_hip_handle *h) {
    if ((int)h_.size() >= nranks()) throw std::invalid_argument("more handles than pieces");
    h_.push_back(h);
    if ((int)h_.size() < nranks()) return;
    int nb = 0;
    for (gpslam_hip_handle *q : h_) {
      int32_t info[4];
      ok(q, gpslam_hip_fs_split_info(q, info), "fs_split_info");
      nb = std::max(nb, (int)info[0]);
    }
    send_.resize(h_.size()); recv_.resize(h_.size());
    for (size_t r = 0; r < h_.size(); r++) {
      hip_ok(hipSetDevice(devices_[r]), "hipSetDevice");
      ok(h_[r], gpslam_hip_set_stream(h_[r], (void *)streams_[r]), "set_stream");
      ok(h_[r], gpslam_hip_fs_set_top(h_[r], nb), "fs_set_top");
      size_t rb = 0;
      ok(h_[r], gpslam_hip_fs_interface(h_[r], &send_[r], &rec_bytes_, &recv_[r], &rb), "fs_interface");
    }
  }
  size_t record_bytes() const { return rec_bytes_; }
  /// one Gauss-Newton (lambda = 0) / damped iteration over all pieces
  gpslam_hip_stats iterate(double lambda = 0.0, bool want_stats = true) {
    if ((int)h_.size() != nranks()) throw std::invalid_argument("register one handle per piece first");
    for (size_t r = 0; r < h_.size(); r++) { use(r); ok(h_[r], gpslam_hip_fs_phase1(h_[r], lambda), "fs_phase1"); }
    gather();
    gpslam_hip_stats tot;
    std::fill((char *)&tot, (char *)&tot + sizeof(tot), 0);
    for (size_t r = 0; r < h_.size(); r++) {
      gpslam_hip_stats st;
      use(r);
      ok(h_[r], gpslam_hip_fs_phase2(h_[r], want_stats ? &st : nullptr), "fs_phase2");
      if (want_stats) {
        tot.error_before += st.error_before;
        tot.error_after += st.error_after;
        tot.delta_inf_norm = std::max(tot.delta_inf_norm, st.delta_inf_norm);
        tot.status = std::min(tot.status, st.status);
      }
    }
    tot.iterations = 1;
    tot.accepted = 1;
    return tot;
  }
  void synchronize() {
    for (size_t r = 0; r < devices_.size(); r++) { use(r); hip_ok(hipStreamSynchronize(streams_[r]), "hipStreamSynchronize"); }
  }

 private:
  void use(size_t r) const { hip_ok(hipSetDevice(devices_[r]), "hipSetDevice"); }
  static void ok(gpslam_hip_handle *h, int rc, const char *what) {
    if (rc < 0) throw std::runtime_error(std::string(what) + " failed: " + gpslam_hip_last_error(h));
  }
  void gather() {
    if (local_) {
      for (size_t r = 0; r < h_.size(); r++)
        for (size_t k = 0; k < h_.size(); k++)
          hip_ok(hipMemcpyAsync((char *)recv_[r] + k * rec_bytes_, send_[k], rec_bytes_, hipMemcpyDeviceToDevice, streams_[0]), "hipMemcpyAsync");
      return;
    }
    nccl_ok(ncclGroupStart(), "ncclGroupStart");
    for (size_t r = 0; r < h_.size(); r++) {
      use(r);
      nccl_ok(ncclAllGather(send_[r], recv_[r], rec_bytes_, ncclChar, comms_[r], streams_[r]), "ncclAllGather");
    }
    nccl_ok(ncclGroupEnd(), "ncclGroupEnd");
  }
  std::vector<int> devices_;
  std::vector<ncclComm_t> comms_;
  std::vector<hipStream_t> streams_;
  std::vector<gpslam_hip_handle *> h_;
  std::vector<void *> send_, recv_;
  size_t rec_bytes_ = 0;
  bool local_ = true;
};

}  // namespace gpslam_hip
