// BASELINE config 4's graph (SE(2) GP chain + odometry + interpolated ranges to locally visible landmarks, matlab/PlazaPose2.m
// :55-66, :147-178 scaled) cut into pieces from C++ (gpslam_amd/host/sharded_host.hpp: SplitDriver) against the unsplit solve.
// One piece per visible GPU over RCCL when the box has several; on a single GPU the same chain is also cut into 3 pieces that
// all live on device 0 (the gather is device copies), so the piece logic runs on the build farm as well.
#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>

#include "../../gpslam_amd/host/sharded_host.hpp"

static void ok(int rc, gpslam_hip_handle *h, const char *what) {
  if (rc < 0) { std::printf("FAILED %s: %s\n", what, h ? gpslam_hip_last_error(h) : ""); std::exit(1); }
}

struct Problem {
  int N, L;
  double dt = 0.1;
  std::vector<double> pose, vel, odo, lmk, lmk_prior;       // N x 3, N x 3, (N - 1) x 3, L x 2, L x 2
  std::vector<int32_t> r_left, r_lm;
  std::vector<double> r_z, r_tau;
};

static Problem make(int N) {
  Problem p;
  p.N = N; p.L = N / 20;
  unsigned s = 777u;
  auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xFFFF) / 65536.0 - 0.5; };
  std::vector<double> truth((size_t)N * 3, 0.0);
  for (int i = 1; i < N; i++) {        // drive along a gently curving path, 0.1 m per step
    const double th = truth[(size_t)(i - 1) * 3 + 2];
    truth[(size_t)i * 3 + 0] = truth[(size_t)(i - 1) * 3 + 0] + 0.1 * std::cos(th);
    truth[(size_t)i * 3 + 1] = truth[(size_t)(i - 1) * 3 + 1] + 0.1 * std::sin(th);
    truth[(size_t)i * 3 + 2] = th + 0.005 * std::sin(0.003 * i);
  }
  p.pose.resize((size_t)N * 3); p.vel.assign((size_t)N * 3, 0.0); p.odo.resize((size_t)(N - 1) * 3);
  for (int i = 0; i < N; i++) {
    p.pose[(size_t)i * 3 + 0] = truth[(size_t)i * 3 + 0] + 0.05 * rnd();
    p.pose[(size_t)i * 3 + 1] = truth[(size_t)i * 3 + 1] + 0.05 * rnd();
    p.pose[(size_t)i * 3 + 2] = truth[(size_t)i * 3 + 2] + 0.01 * rnd();
    p.vel[(size_t)i * 3 + 0] = 1.0;
  }
  for (int i = 0; i + 1 < N; i++) {    // relative pose in the frame of state i
    const double *a = &truth[(size_t)i * 3], *b = &truth[(size_t)(i + 1) * 3];
    const double c = std::cos(a[2]), sn = std::sin(a[2]), dx = b[0] - a[0], dy = b[1] - a[1];
    p.odo[(size_t)i * 3 + 0] = c * dx + sn * dy + 1e-3 * rnd();
    p.odo[(size_t)i * 3 + 1] = -sn * dx + c * dy + 1e-3 * rnd();
    p.odo[(size_t)i * 3 + 2] = b[2] - a[2] + 1e-3 * rnd();
  }
  p.lmk.resize((size_t)p.L * 2); p.lmk_prior.resize((size_t)p.L * 2);
  std::vector<double> lt((size_t)p.L * 2);
  for (int l = 0; l < p.L; l++) {
    const int c = std::min(N - 1, l * 20 + 10);
    const double th = truth[(size_t)c * 3 + 2], side = (l & 1) ? 8.0 : -11.0;
    lt[(size_t)l * 2 + 0] = truth[(size_t)c * 3 + 0] - side * std::sin(th);
    lt[(size_t)l * 2 + 1] = truth[(size_t)c * 3 + 1] + side * std::cos(th);
    for (int q = 0; q < 2; q++) {
      p.lmk_prior[(size_t)l * 2 + q] = lt[(size_t)l * 2 + q];
      p.lmk[(size_t)l * 2 + q] = lt[(size_t)l * 2 + q] + 0.4 * rnd();
    }
  }
  for (int i = 0; i + 1 < N; i++) {
    if (rnd() > -0.06) continue;                      // ~0.44 range factors per interval
    const int lc = i / 20, l = std::min(p.L - 1, std::max(0, lc + (int)std::floor(8.0 * rnd())));   // closest approach within ~100 states
    const double tau = p.dt * (rnd() + 0.5);
    const double a = tau / p.dt, x = (1 - a) * truth[(size_t)i * 3] + a * truth[(size_t)(i + 1) * 3],
                 y = (1 - a) * truth[(size_t)i * 3 + 1] + a * truth[(size_t)(i + 1) * 3 + 1];
    p.r_left.push_back(i); p.r_lm.push_back(l); p.r_tau.push_back(tau);
    p.r_z.push_back(std::hypot(lt[(size_t)l * 2] - x, lt[(size_t)l * 2 + 1] - y) + 0.05 * rnd());
  }
  return p;
}

struct Piece {
  gpslam_hip_handle *h = nullptr;
  int lo = 0, hi = 0;
  std::vector<int> lm_global;        // local landmark -> global
  std::vector<char> own;             // this piece reports the landmark
};

// states [lo, hi] (both ends); factors with left state in [lo, hi) -- the last piece also those of state hi; the landmarks
// those range factors touch; a landmark seen from two pieces goes to both (prior to the right one)
static Piece build(const Problem &p, int device, int rank, int P, const std::vector<int> &bounds, bool split) {
  Piece pc;
  pc.lo = bounds[rank]; pc.hi = bounds[rank + 1];
  const int lo = pc.lo, hi = pc.hi, n = hi - lo + 1;
  const bool last = rank == P - 1;
  auto mine = [&](int i) { return i >= lo && (i < hi || (last && i == hi)); };
  auto piece_of = [&](int i) { int r = 0; while (r + 1 < P && i >= bounds[r + 1]) r++; return r; };
  std::vector<int> pmin(p.L, P), pmax(p.L, -1);
  for (size_t k = 0; k < p.r_left.size(); k++) {
    const int r = piece_of(p.r_left[k]);
    pmin[p.r_lm[k]] = std::min(pmin[p.r_lm[k]], r);
    pmax[p.r_lm[k]] = std::max(pmax[p.r_lm[k]], r);
  }
  std::vector<int> g2l(p.L, -1);
  std::vector<int32_t> first, lastl;
  std::vector<double> lm;
  for (int l = 0; l < p.L; l++) {
    if (pmax[l] < 0) { pmin[l] = pmax[l] = 0; }
    if (pmax[l] - pmin[l] > 1) { std::printf("FAILED: a landmark is seen from three pieces\n"); std::exit(1); }
    if (pmin[l] > rank || pmax[l] < rank) continue;
    g2l[l] = (int)pc.lm_global.size();
    if (pmin[l] < rank) first.push_back(g2l[l]);
    if (pmax[l] > rank) lastl.push_back(g2l[l]);
    pc.lm_global.push_back(l);
    pc.own.push_back(pmax[l] == rank);
    lm.push_back(p.lmk[(size_t)l * 2]); lm.push_back(p.lmk[(size_t)l * 2 + 1]);
  }
  gpslam_hip_config_v2 cfg;
  std::memset(&cfg, 0, sizeof(cfg));
  cfg.struct_size = (uint32_t)sizeof(cfg);
  cfg.manifold = GPSLAM_POSE2; cfg.precision = GPSLAM_FP64; cfg.device = device; cfg.rank = 0; cfg.nranks = 1;
  cfg.chart = GPSLAM_CHART_FIRST_ORDER; cfg.landmark_dim = 2;
  cfg.force_segmented = 1;             // the reference run takes the segmented path as well
  ok(gpslam_hip_create_v2(&cfg, &pc.h) != 0 ? -1 : 0, nullptr, "create");
  gpslam_hip_handle *h = pc.h;
  ok(gpslam_hip_set_states(h, n, &p.pose[(size_t)lo * 3], &p.vel[(size_t)lo * 3]), h, "set_states");
  ok(gpslam_hip_set_landmarks(h, (int)pc.lm_global.size(), lm.data()), h, "set_landmarks");
  const double Qc[9] = {0.01, 0, 0, 0, 0.01, 0, 0, 0, 0.01};
  ok(gpslam_hip_set_qc(h, Qc), h, "set_qc");
  if (split) ok(gpslam_hip_fs_set_split(h, rank, P, first.data(), (int)first.size(), lastl.data(), (int)lastl.size()), h, "fs_set_split");
  std::vector<int32_t> left;
  std::vector<double> dts, om, os;
  for (int i = lo; i < hi; i++) {
    left.push_back(i - lo); dts.push_back(p.dt);
    for (int q = 0; q < 3; q++) om.push_back(p.odo[(size_t)i * 3 + q]);
    os.push_back(1e-3); os.push_back(1e-3); os.push_back(3e-3);
  }
  ok(gpslam_hip_add_gp_priors(h, (int)left.size(), left.data(), dts.data()), h, "add_gp_priors");
  ok(gpslam_hip_add_between(h, (int)left.size(), left.data(), om.data(), os.data()), h, "add_between");
  if (mine(0)) {
    const int32_t z = 0;
    const double sg[3] = {0.1, 0.1, 0.05};
    ok(gpslam_hip_add_pose_priors(h, 1, &z, &p.pose[0], sg), h, "add_pose_priors");
  }
  std::vector<int32_t> pi;
  std::vector<double> pm, ps;
  for (size_t l = 0; l < pc.lm_global.size(); l++)
    if (pc.own[l]) {
      pi.push_back((int32_t)l);
      for (int q = 0; q < 2; q++) { pm.push_back(p.lmk_prior[(size_t)pc.lm_global[l] * 2 + q]); ps.push_back(1.0); }
    }
  ok(gpslam_hip_add_landmark_priors(h, (int)pi.size(), pi.data(), pm.data(), ps.data()), h, "add_landmark_priors");
  std::vector<int32_t> rl, rm;
  std::vector<double> rz, rs, rd, rt;
  for (size_t k = 0; k < p.r_left.size(); k++)
    if (mine(p.r_left[k])) {
      rl.push_back(p.r_left[k] - lo); rm.push_back(g2l[p.r_lm[k]]);
      rz.push_back(p.r_z[k]); rs.push_back(0.5); rd.push_back(p.dt); rt.push_back(p.r_tau[k]);
    }
  ok(gpslam_hip_add_interp_range(h, (int)rl.size(), rl.data(), rm.data(), rz.data(), rs.data(), rd.data(), rt.data(), nullptr), h, "add_interp_range");
  ok(gpslam_hip_compile(h), h, "compile");
  return pc;
}

static bool run(const Problem &p, const std::vector<int> &devs, const std::vector<double> &xr, const std::vector<double> &vr,
                const std::vector<double> &lr, double err_ref) {
  const int P = (int)devs.size(), N = p.N;
  std::vector<int> bounds(P + 1);
  for (int r = 0; r <= P; r++) bounds[r] = (int)((long)r * (N - 1) / P);
  gpslam_hip_stats st;
  double worst = 0.0;
  {
    gpslam_hip::SplitDriver drv(devs);
    std::vector<Piece> pcs;
    for (int r = 0; r < P; r++) {
      pcs.push_back(build(p, devs[r], r, P, bounds, true));
      drv.add(pcs.back().h);
    }
    for (int it = 0; it < 5; it++) st = drv.iterate(0.0, true);
    drv.synchronize();
    for (const Piece &pc : pcs) {
      const int n = pc.hi - pc.lo + 1;
      std::vector<double> x((size_t)n * 3), v((size_t)n * 3), lm(pc.lm_global.size() * 2);
      ok(gpslam_hip_get_states(pc.h, x.data(), v.data()), pc.h, "get_states");
      ok(gpslam_hip_get_landmarks(pc.h, lm.data()), pc.h, "get_landmarks");
      for (size_t k = 0; k < x.size(); k++) {
        worst = std::fmax(worst, std::fabs(x[k] - xr[(size_t)pc.lo * 3 + k]));
        worst = std::fmax(worst, std::fabs(v[k] - vr[(size_t)pc.lo * 3 + k]));
      }
      for (size_t l = 0; l < pc.lm_global.size(); l++)
        for (int q = 0; q < 2; q++) worst = std::fmax(worst, std::fabs(lm[l * 2 + q] - lr[(size_t)pc.lm_global[l] * 2 + q]));
    }
    std::printf("pieces %d (%s), states %d, landmarks %d, record %zu bytes: error %.9e (unsplit %.9e), max |difference| %.3e\n", P,
                P > 1 && devs[0] == devs[1] ? "one device, device copies" : "RCCL", N, p.L, drv.record_bytes(), st.error_after, err_ref, worst);
    for (Piece &pc : pcs) gpslam_hip_destroy(pc.h);
  }
  return worst <= 1e-8 && std::fabs(st.error_after - err_ref) <= 1e-8 * std::fmax(1.0, err_ref);
}

int main() {
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) { std::printf("FAILED: no HIP device\n"); return 1; }
  const int Pd = std::min(ndev, 8);
  const int N = 3000 * std::max(Pd, 3);
  const Problem p = make(N);
  // reference: the whole chain on one device, unsplit
  std::vector<int> whole = {0, N - 1};
  Piece ref = build(p, 0, 0, 1, whole, false);
  gpslam_hip_stats st_ref;
  for (int it = 0; it < 5; it++) ok(gpslam_hip_iterate_gn(ref.h, &st_ref), ref.h, "iterate_gn");
  std::vector<double> xr((size_t)N * 3), vr((size_t)N * 3), lr((size_t)p.L * 2);
  ok(gpslam_hip_get_states(ref.h, xr.data(), vr.data()), ref.h, "get_states");
  ok(gpslam_hip_get_landmarks(ref.h, lr.data()), ref.h, "get_landmarks");
  gpslam_hip_destroy(ref.h);
  bool pass = true;
  std::vector<int> devs(Pd);
  for (int r = 0; r < Pd; r++) devs[r] = r;
  pass = run(p, devs, xr, vr, lr, st_ref.error_after) && pass;                  // one piece per GPU, RCCL
  pass = run(p, std::vector<int>(3, 0), xr, vr, lr, st_ref.error_after) && pass;   // three pieces on device 0
  std::printf(pass ? "split_rccl_test: all tests passed\n" : "split_rccl_test: FAILED\n");
  return pass ? 0 : 1;
}
