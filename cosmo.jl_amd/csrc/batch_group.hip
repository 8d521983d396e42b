// batch_group.hip -- batches of independent problems of DIFFERENT structure (round 5).
//
// The reference's batch mode is a loop `for model in models; optimize!(model); end` over arbitrary models (src/solver.jl:78): nothing ties the
// problems of a batch to one shape.  The persistent kernels of batch.hip are specialised per structure -- (n, m, cone table) choose the kernel form,
// its LDS image layout and its register arrays at set_params time -- and that specialisation is where their speed comes from.  A heterogeneous batch
// therefore keeps it: the problems are partitioned into CLASSES of identical structure (dimensions, cone types, cone dimensions, cone parameters; the
// data, Box bounds and scalings differ freely), every class becomes one cosmo_hip_batch with its own HIP stream, and cosmo_hip_batch_group_optimize
// drives all classes CONCURRENTLY, one host thread per class (the per-class host loop of cosmo_hip_batch_optimize -- persistent launch, certificate
// kernels, status copy -- synchronises only its own stream), so that the workgroups of all classes share the chip like those of one batch do.  A class
// may hold a single problem (one persistent workgroup).  Results, iterates and counters are addressed by the caller's problem index.
// A class whose structure the batch kernels REFUSE (cosmo_hip_batch_* returns COSMO_HIP_ERR_UNSUPPORTED: a PSD cone of side > 64, a KKT solver kind other
// than the CG kinds, the automatic rho interval) is not an error of the group: its members are solved through one single-problem handle each
// (cosmo_hip_create ... cosmo_hip_optimize: the device-resident loop of api.hip with the batched matrix-sign projections), concurrently with the batch
// classes -- so a group accepts every problem the library can solve at all, as the reference's loop over models does.
//
// Host-side orchestration only: no kernel lives here.  Replaces: the reference's loop over models, src/solver.jl:78-203 per model.
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <atomic>
#include <map>
#include <string>
#include <thread>
#include <vector>
#include "internal.h"

namespace {

struct GProblem {
  long long n = 0, m = 0;
  std::vector<int64_t> Pp, Pi, Ap, Ai;
  std::vector<real> Px, Ax, q, b, box_l, box_u, Dinv, Einv, Dsc, Esc, x0, s0, mu0;     // Dsc / Esc: D, E themselves where the caller handed them over (set_scaling_full)
  std::vector<int32_t> ctype; std::vector<int64_t> cdim; std::vector<real> cparam;
  double cinv = 1.0;
  bool have = false, have_cones = false, have_scaling = false, have_x0 = false, have_s0 = false, have_mu0 = false;
  int cls = -1, pos = -1;         // class and position inside the class
  bool dirty = true;              // set_iterates was called for THIS problem since the class last ran (or it never ran)
};

struct GClass {
  cosmo_hip_batch* b = nullptr;    // the class's persistent-kernel batch ...
  std::vector<cosmo_hip_handle*> hs;   // ... or, for a structure the batch kernels do not take (a PSD cone of side > 64, a MINRES solver kind), one
                                       // single-problem handle per member (the per-problem path of api.hip: the batched matrix-sign projections etc.)
  std::vector<int> members;       // problem indices, ascending
  long long n = 0, m = 0, nbox = 0;
  bool iterates_dirty = true;
  bool ran = false;               // the class has been optimized at least once (its device state is a solved / warm state)
};

}  // namespace

struct cosmo_hip_batch_group {
  int device = 0;
  std::string err;
  std::vector<GProblem> prob;
  std::vector<GClass> cls;
  bool finalized = false, aa_on = false;
  int last_workers = 0; long long last_jobs = 0, last_merged = 0;     // of the last optimize: worker threads, jobs (merged sets + batch classes + members on their own handles), classes in merged sets
  cosmo_hip_accel_params aa;
  cosmo_hip_params prm;
};

static int32_t gfail(cosmo_hip_batch_group* g, int32_t code, const std::string& msg) { if (g) g->err = msg; return code; }
#define GCHECK(g, k) do { if (!(g)) return COSMO_HIP_ERR_INVALID; if ((k) < 0 || (k) >= (int64_t)(g)->prob.size()) return gfail((g), COSMO_HIP_ERR_INVALID, "problem index out of range"); } while (0)

extern "C" int32_t cosmo_hip_batch_group_create(cosmo_hip_batch_group** out, int32_t device_id, int64_t nprob) {
  if (!out || nprob <= 0) return COSMO_HIP_ERR_INVALID;
  *out = nullptr;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return COSMO_HIP_ERR_HIP;        // no CPU fallback
  if (device_id < 0 || device_id >= ndev) return COSMO_HIP_ERR_INVALID;
  cosmo_hip_batch_group* g = new cosmo_hip_batch_group();
  g->device = device_id;
  g->prob.resize((size_t)nprob);
  cosmo_hip_default_params(&g->prm);
  memset(&g->aa, 0, sizeof g->aa);
  *out = g;
  return COSMO_HIP_OK;
}

extern "C" int32_t cosmo_hip_batch_group_destroy(cosmo_hip_batch_group* g) {
  if (!g) return COSMO_HIP_OK;
  for (auto& c : g->cls) { if (c.b) (void)cosmo_hip_batch_destroy(c.b); for (auto* h : c.hs) if (h) (void)cosmo_hip_destroy(h); }
  delete g;
  return COSMO_HIP_OK;
}

extern "C" const char* cosmo_hip_batch_group_last_error(const cosmo_hip_batch_group* g) { return g ? g->err.c_str() : "null batch group"; }

// problem k with ITS OWN dimensions; arrays as cosmo_hip_set_problem (Julia's SparseMatrixCSC: 1-based colptr / rowval)
extern "C" int32_t cosmo_hip_batch_group_set_problem(cosmo_hip_batch_group* g, int64_t k, int64_t n, int64_t m, const int64_t* P_colptr, const int64_t* P_rowval,
                                                     const real* P_nzval, const int64_t* A_colptr, const int64_t* A_rowval, const real* A_nzval, const real* q,
                                                     const real* bvec) {
  GCHECK(g, k);
  if (g->finalized) return gfail(g, COSMO_HIP_ERR_INVALID, "batch_group_set_problem: after set_params");
  if (n < 0 || m < 0 || !P_colptr || !A_colptr || (n > 0 && !q) || (m > 0 && !bvec)) return gfail(g, COSMO_HIP_ERR_INVALID, "batch_group_set_problem: bad arguments");
  if (P_colptr[0] != 1 || A_colptr[0] != 1) return gfail(g, COSMO_HIP_ERR_INVALID, "colptr must be 1-based");
  GProblem& p = g->prob[(size_t)k];
  p.n = n; p.m = m;
  const int64_t nnzP = P_colptr[n] - 1, nnzA = A_colptr[n] - 1;
  if (nnzP < 0 || nnzA < 0) return gfail(g, COSMO_HIP_ERR_INVALID, "batch_group_set_problem: bad colptr");
  p.Pp.assign(P_colptr, P_colptr + n + 1); p.Pi.assign(P_rowval, P_rowval + nnzP); p.Px.assign(P_nzval, P_nzval + nnzP);
  p.Ap.assign(A_colptr, A_colptr + n + 1); p.Ai.assign(A_rowval, A_rowval + nnzA); p.Ax.assign(A_nzval, A_nzval + nnzA);
  p.q.assign(q, q + n); p.b.assign(bvec, bvec + m);
  p.have = true;
  return COSMO_HIP_OK;
}

// cones of problem k (as cosmo_hip_set_cones_ex; box_l / box_u: the Box rows of THIS problem, in order)
extern "C" int32_t cosmo_hip_batch_group_set_cones(cosmo_hip_batch_group* g, int64_t k, int64_t ncones, const int32_t* type, const int64_t* dim,
                                                   const real* box_l, const real* box_u, const real* cone_param) {
  GCHECK(g, k);
  if (g->finalized) return gfail(g, COSMO_HIP_ERR_INVALID, "batch_group_set_cones: after set_params");
  if (ncones < 0 || (ncones > 0 && (!type || !dim))) return gfail(g, COSMO_HIP_ERR_INVALID, "batch_group_set_cones: bad arguments");
  GProblem& p = g->prob[(size_t)k];
  p.ctype.assign(type, type + ncones); p.cdim.assign(dim, dim + ncones);
  p.cparam.assign((size_t)ncones, R(0.0));
  long long nbox = 0;
  for (int64_t c = 0; c < ncones; ++c) {
    if (cone_param && (type[c] == COSMO_HIP_POW || type[c] == COSMO_HIP_DUAL_POW)) p.cparam[(size_t)c] = cone_param[c];
    if (type[c] == COSMO_HIP_BOX) nbox += dim[c];
  }
  if (nbox > 0 && (!box_l || !box_u)) return gfail(g, COSMO_HIP_ERR_INVALID, "batch_group_set_cones: Box cones need box_l / box_u");
  p.box_l.assign(box_l, box_l + nbox); p.box_u.assign(box_u, box_u + nbox);
  p.have_cones = true;
  return COSMO_HIP_OK;
}

extern "C" int32_t cosmo_hip_batch_group_set_scaling(cosmo_hip_batch_group* g, int64_t k, const real* Dinv, const real* Einv, double cinv) {
  GCHECK(g, k);
  if (g->finalized) return gfail(g, COSMO_HIP_ERR_INVALID, "batch_group_set_scaling: after set_params");
  GProblem& p = g->prob[(size_t)k];
  if (!p.have) return gfail(g, COSMO_HIP_ERR_INVALID, "batch_group_set_scaling: set_problem first");
  if (Dinv) p.Dinv.assign(Dinv, Dinv + p.n); else p.Dinv.assign((size_t)p.n, R(1.0));
  if (Einv) p.Einv.assign(Einv, Einv + p.m); else p.Einv.assign((size_t)p.m, R(1.0));
  p.cinv = cinv; p.have_scaling = true;
  return COSMO_HIP_OK;
}

// The same with D, E and c themselves (cosmo_hip_set_scaling_full): a member that runs on its own handle then carries EXACTLY the scaling matrices of a
// single-handle solve into its infeasibility tests (src/infeasibility.jl:5,35,39) instead of reciprocals of reciprocals.
extern "C" int32_t cosmo_hip_batch_group_set_scaling_full(cosmo_hip_batch_group* g, int64_t k, const real* D, const real* Dinv, const real* E, const real* Einv, double c,
                                                          double cinv) {
  const int32_t rc = cosmo_hip_batch_group_set_scaling(g, k, Dinv, Einv, cinv);
  if (rc) return rc;
  (void)c;
  GProblem& p = g->prob[(size_t)k];
  if (D) p.Dsc.assign(D, D + p.n); else p.Dsc.clear();
  if (E) p.Esc.assign(E, E + p.m); else p.Esc.clear();
  return COSMO_HIP_OK;
}

// the reference's accelerator for every problem (as cosmo_hip_batch_set_accelerator); before set_params
extern "C" int32_t cosmo_hip_batch_group_set_accelerator(cosmo_hip_batch_group* g, const cosmo_hip_accel_params* p) {
  if (!g) return COSMO_HIP_ERR_INVALID;
  if (g->finalized) return gfail(g, COSMO_HIP_ERR_INVALID, "batch_group_set_accelerator: after set_params");
  g->aa_on = p && p->kind != COSMO_HIP_ACCEL_EMPTY;
  if (p) g->aa = *p;
  return COSMO_HIP_OK;
}

// Partitions the problems into classes of identical structure and finalises one cosmo_hip_batch per class -- or, where the batch kernels refuse the
// structure (COSMO_HIP_ERR_UNSUPPORTED), one single-problem handle per member.  Any other error of a class's set-up, and a member that no path of the
// library takes, is the group's error with the offending problem named; the group then holds no classes and set_params may be called again.
extern "C" int32_t cosmo_hip_batch_group_set_params(cosmo_hip_batch_group* g, const cosmo_hip_params* prm) {
  if (!g || !prm) return COSMO_HIP_ERR_INVALID;
  if (g->finalized) return gfail(g, COSMO_HIP_ERR_INVALID, "batch_group_set_params: called twice");
  g->prm = *prm;
  for (auto& c : g->cls) { if (c.b) (void)cosmo_hip_batch_destroy(c.b); for (auto* h : c.hs) if (h) (void)cosmo_hip_destroy(h); }     // (left over from a failed call)
  g->cls.clear();
  std::map<std::string, int> index;
  for (size_t k = 0; k < g->prob.size(); ++k) {
    GProblem& p = g->prob[k];
    if (!p.have || !p.have_cones) return gfail(g, COSMO_HIP_ERR_INVALID, "batch_group_set_params: problem " + std::to_string(k) + " has no data / no cones");
    std::string key;
    auto put = [&](const void* src, size_t bytes) { key.append(reinterpret_cast<const char*>(src), bytes); };
    put(&p.n, sizeof p.n); put(&p.m, sizeof p.m);
    const size_t nc = p.ctype.size(); put(&nc, sizeof nc);
    put(p.ctype.data(), nc * sizeof(int32_t)); put(p.cdim.data(), nc * sizeof(int64_t)); put(p.cparam.data(), nc * sizeof(real));
    auto it = index.find(key);
    if (it == index.end()) { it = index.emplace(key, (int)g->cls.size()).first; g->cls.emplace_back(); g->cls.back().n = p.n; g->cls.back().m = p.m; }
    p.cls = it->second; p.pos = (int)g->cls[(size_t)it->second].members.size();
    g->cls[(size_t)it->second].members.push_back((int)k);
  }
  for (size_t ci = 0; ci < g->cls.size(); ++ci) {
    GClass& C = g->cls[ci];
    const GProblem& p0 = g->prob[(size_t)C.members[0]];
    auto bad = [&](int32_t rc, const char* what) {
      const std::string detail = C.b ? cosmo_hip_batch_last_error(C.b) : "";
      return gfail(g, rc, std::string(what) + " failed for the class of problem " + std::to_string(C.members[0]) + " (" + std::to_string(C.members.size()) +
                          " problem(s), n = " + std::to_string(C.n) + ", m = " + std::to_string(C.m) + "): " + detail);
    };
    int32_t rc = cosmo_hip_batch_create(&C.b, g->device, (int64_t)C.members.size(), C.n, C.m);
    if (rc) return bad(rc, "batch_create");
    long long nbox = 0;
    for (size_t c = 0; c < p0.ctype.size(); ++c) if (p0.ctype[c] == COSMO_HIP_BOX) nbox += p0.cdim[c];
    C.nbox = nbox;
    std::vector<real> bl((size_t)(nbox * (long long)C.members.size())), bu(bl.size());
    for (size_t j = 0; j < C.members.size(); ++j) {
      const GProblem& p = g->prob[(size_t)C.members[j]];
      rc = cosmo_hip_batch_set_problem(C.b, (int64_t)j, p.Pp.data(), p.Pi.data(), p.Px.data(), p.Ap.data(), p.Ai.data(), p.Ax.data(), p.q.data(), p.b.data());
      if (rc) return bad(rc, "batch_set_problem");
      if (nbox) { std::copy(p.box_l.begin(), p.box_l.end(), bl.begin() + (size_t)(nbox * (long long)j)); std::copy(p.box_u.begin(), p.box_u.end(), bu.begin() + (size_t)(nbox * (long long)j)); }
      if (p.have_scaling) { rc = cosmo_hip_batch_set_scaling(C.b, (int64_t)j, p.Dinv.data(), p.Einv.data(), p.cinv); if (rc) return bad(rc, "batch_set_scaling"); }
    }
    rc = cosmo_hip_batch_set_cones_ex(C.b, (int64_t)p0.ctype.size(), p0.ctype.data(), p0.cdim.data(), bl.data(), bu.data(), p0.cparam.data());
    if (rc == COSMO_HIP_OK && g->aa_on) rc = cosmo_hip_batch_set_accelerator(C.b, &g->aa);
    if (rc == COSMO_HIP_OK) rc = cosmo_hip_batch_set_params(C.b, prm);
    if (rc == COSMO_HIP_ERR_UNSUPPORTED) {
      // not a structure of the persistent kernels: one single-problem handle per member instead
      const std::string why = cosmo_hip_batch_last_error(C.b);
      (void)cosmo_hip_batch_destroy(C.b); C.b = nullptr;
      auto hbad = [&](int32_t code, cosmo_hip_handle* h, const char* what, int k) {
        const std::string detail = h ? cosmo_hip_last_error(h) : "";
        return gfail(g, code, std::string(what) + " failed for problem " + std::to_string(k) + " (solved through its own handle because the batch kernels refuse its structure: " + why + "): " + detail);
      };
      for (size_t j = 0; j < C.members.size(); ++j) {
        const int k = C.members[j];
        const GProblem& p = g->prob[(size_t)k];
        cosmo_hip_handle* h = nullptr;
        int32_t hr = cosmo_hip_create(&h, g->device);
        if (hr) return hbad(hr, nullptr, "cosmo_hip_create", k);
        C.hs.push_back(h);
        if ((hr = cosmo_hip_set_problem(h, p.n, p.m, p.Pp.data(), p.Pi.data(), p.Px.data(), p.Ap.data(), p.Ai.data(), p.Ax.data(), p.q.data(), p.b.data()))) return hbad(hr, h, "set_problem", k);
        if ((hr = cosmo_hip_set_cones_ex(h, (int64_t)p.ctype.size(), p.ctype.data(), p.cdim.data(), p.box_l.empty() ? nullptr : p.box_l.data(),
                                         p.box_u.empty() ? nullptr : p.box_u.data(), p.cparam.data()))) return hbad(hr, h, "set_cones", k);
        if ((hr = cosmo_hip_set_params(h, prm, nullptr))) return hbad(hr, h, "set_params", k);
        if (p.have_scaling) {
          if (!p.Dsc.empty() && !p.Esc.empty()) hr = cosmo_hip_set_scaling_full(h, p.Dsc.data(), p.Dinv.data(), p.Esc.data(), p.Einv.data(), 1.0 / p.cinv, p.cinv);
          else hr = cosmo_hip_set_scaling(h, p.Dinv.data(), p.Einv.data(), p.cinv);
          if (hr) return hbad(hr, h, "set_scaling", k);
        }
        if (g->aa_on && (hr = cosmo_hip_set_accelerator(h, &g->aa))) return hbad(hr, h, "set_accelerator", k);      // (the order of optimize_hip! / model.setup)
      }
      continue;
    }
    if (rc) return bad(rc, "batch set-up");
  }
  for (auto& p : g->prob) {            // the staged matrices are on the device now
    std::vector<int64_t>().swap(p.Pp); std::vector<int64_t>().swap(p.Pi); std::vector<int64_t>().swap(p.Ap); std::vector<int64_t>().swap(p.Ai);
    std::vector<real>().swap(p.Px); std::vector<real>().swap(p.Ax);
  }
  g->finalized = true;
  return COSMO_HIP_OK;
}

// out_nclasses: number of structure classes; class_of[k] (nprob entries, may be NULL): class of problem k
extern "C" int32_t cosmo_hip_batch_group_class_info(cosmo_hip_batch_group* g, int64_t* out_nclasses, int64_t* class_of, int64_t* mode_of) {
  if (!g || !g->finalized) return gfail(g, COSMO_HIP_ERR_INVALID, "batch_group_class_info: set_params first");
  if (out_nclasses) *out_nclasses = (int64_t)g->cls.size();
  if (class_of) for (size_t k = 0; k < g->prob.size(); ++k) class_of[k] = g->prob[k].cls;
  if (mode_of) for (size_t k = 0; k < g->prob.size(); ++k) mode_of[k] = g->cls[(size_t)g->prob[k].cls].b ? 0 : 1;     // 0: persistent batch kernel, 1: its own handle
  return COSMO_HIP_OK;
}

// the adaptive-rho interval in force for problem k: out = {interval (0: the automatic rule of solver.jl:244-256 has not fired), iteration at which the
// automatic rule fixed it or -1} -- what the reference writes back into settings.adaptive_rho_interval (solver.jl:249-254).  Problems inside a batch class
// run on the fixed interval of the group's parameters (the automatic interval is one of the structures the batch kernels refuse).
extern "C" int32_t cosmo_hip_batch_group_get_rho_interval(cosmo_hip_batch_group* g, int64_t k, int64_t out[2]) {
  GCHECK(g, k);
  if (!g->finalized || !out) return gfail(g, COSMO_HIP_ERR_INVALID, "batch_group_get_rho_interval: set_params first");
  const GProblem& p = g->prob[(size_t)k];
  GClass& C = g->cls[(size_t)p.cls];
  if (!C.b) { const int32_t rc = cosmo_hip_get_rho_interval(C.hs[(size_t)p.pos], out); return rc ? gfail(g, rc, cosmo_hip_last_error(C.hs[(size_t)p.pos])) : COSMO_HIP_OK; }
  out[0] = g->prm.adaptive_rho_interval; out[1] = -1;
  return COSMO_HIP_OK;
}

// out = {worker threads of the last optimize, its jobs (merged sets + batch classes + members solved on their own handles), classes, problems, classes that
// ran inside a merged set}
extern "C" int32_t cosmo_hip_batch_group_run_info(cosmo_hip_batch_group* g, int64_t out[5]) {
  if (!g || !out) return COSMO_HIP_ERR_INVALID;
  out[0] = g->last_workers; out[1] = g->last_jobs; out[2] = (int64_t)g->cls.size(); out[3] = (int64_t)g->prob.size(); out[4] = g->last_merged;
  return COSMO_HIP_OK;
}

// warm start of problem k (NULL = zeros; src/solver.jl:128-129); problems never set start from zero
extern "C" int32_t cosmo_hip_batch_group_set_iterates(cosmo_hip_batch_group* g, int64_t k, const real* x0, const real* s0, const real* mu0) {
  GCHECK(g, k);
  if (!g->finalized) return gfail(g, COSMO_HIP_ERR_INVALID, "batch_group_set_iterates: set_params first");
  GProblem& p = g->prob[(size_t)k];
  p.have_x0 = x0 != nullptr; p.have_s0 = s0 != nullptr; p.have_mu0 = mu0 != nullptr;
  if (x0) p.x0.assign(x0, x0 + p.n);
  if (s0) p.s0.assign(s0, s0 + p.m);
  if (mu0) p.mu0.assign(mu0, mu0 + p.m);
  p.dirty = true;
  g->cls[(size_t)p.cls].iterates_dirty = true;
  return COSMO_HIP_OK;
}

static int32_t flush_iterates(cosmo_hip_batch_group* g, GClass& C) {
  if (!C.iterates_dirty) return COSMO_HIP_OK;
  const size_t np = C.members.size();
  // Dirtiness is tracked PER PROBLEM: a set_iterates on one member after the class has run must not throw the others back to their old
  // staged start.  Single-handle members: only the touched handles are reset.  Batch classes: cosmo_hip_batch_set_iterates restarts the whole class, so
  // the untouched members are handed their CURRENT device state (x, s, mu) as their start -- the warm start they already had.
  if (!C.b) {
    for (size_t j = 0; j < np; ++j) {
      GProblem& p = g->prob[(size_t)C.members[j]];
      if (C.ran && !p.dirty) continue;
      const int32_t rc = cosmo_hip_set_iterates(C.hs[j], p.have_x0 ? p.x0.data() : nullptr, p.have_s0 ? p.s0.data() : nullptr, p.have_mu0 ? p.mu0.data() : nullptr);
      if (rc) return gfail(g, rc, std::string("set_iterates of problem ") + std::to_string(C.members[j]) + ": " + cosmo_hip_last_error(C.hs[j]));
      p.dirty = false;
    }
    C.iterates_dirty = false;
    return COSMO_HIP_OK;
  }
  std::vector<real> x((size_t)C.n * np, R(0.0)), s((size_t)C.m * np, R(0.0)), mu((size_t)C.m * np, R(0.0));
  std::vector<real> wk((size_t)(C.n + C.m)), wp((size_t)(C.n + C.m));
  for (size_t j = 0; j < np; ++j) {
    GProblem& p = g->prob[(size_t)C.members[j]];
    if (C.ran && !p.dirty) {
      const int32_t rc = cosmo_hip_batch_get_iterates(C.b, (int64_t)j, wk.data(), wp.data(), s.data() + (size_t)C.m * j, mu.data() + (size_t)C.m * j);
      if (rc) return gfail(g, rc, std::string("batch_get_iterates: ") + cosmo_hip_batch_last_error(C.b));
      std::copy(wp.begin(), wp.begin() + C.n, x.begin() + (size_t)C.n * j);       // x is the head of w_prev (src/types.jl:274)
      continue;
    }
    if (p.have_x0) std::copy(p.x0.begin(), p.x0.end(), x.begin() + (size_t)C.n * j);
    if (p.have_s0) std::copy(p.s0.begin(), p.s0.end(), s.begin() + (size_t)C.m * j);
    if (p.have_mu0) std::copy(p.mu0.begin(), p.mu0.end(), mu.begin() + (size_t)C.m * j);
    p.dirty = false;
  }
  const int32_t rc = cosmo_hip_batch_set_iterates(C.b, x.data(), s.data(), mu.data());
  if (rc) return gfail(g, rc, std::string("batch_set_iterates: ") + cosmo_hip_batch_last_error(C.b));
  C.iterates_dirty = false;
  return COSMO_HIP_OK;
}

// optimize! for every problem; results has nprob entries (the caller's order).  The classes run concurrently.
extern "C" int32_t cosmo_hip_batch_group_optimize(cosmo_hip_batch_group* g, cosmo_hip_result* results) {
  if (!g || !g->finalized || !results) return gfail(g, COSMO_HIP_ERR_INVALID, "batch_group_optimize: set_params first");
  for (auto& C : g->cls) { const int32_t rc = flush_iterates(g, C); if (rc) return rc; }
  const size_t nc = g->cls.size();
  std::vector<std::vector<cosmo_hip_result>> res(nc);
  std::vector<int32_t> rcs(nc, COSMO_HIP_OK);
  // JOBS: one per batch class (the class's own host loop on its own stream: persistent launch, certificate kernels, status copy) and one per MEMBER of a
  // class that runs on single-problem handles (each handle has its own stream, so refused members overlap like classes do -- until round 6 they ran one
  // after the other inside their class's thread).  A BOUNDED pool of worker threads takes the jobs off a shared counter, largest first: a list of 1024
  // models of 1024 shapes (the reference's `for model in models; optimize!(model)`, src/solver.jl:78) is 1024 jobs on at most COSMO_HIP_GROUP_WORKERS
  // (default 32) threads and streams in flight, not 1024 threads -- the chip has 256 CUs, and the runtime serialises the submissions anyway.
  // MERGED SETS (round 6): the singleton classes whose batch the streaming kernel takes (batch_multi_supported) do not get a host loop each -- all of them run
  // in ONE host loop whose launches cover every member (batch.hip: batch_multi_optimize, k_batch_admm_multi: workgroup c reads the descriptor of batch c) --
  // one set for the classes that need the extended-cone code, one for the others.  COSMO_HIP_GROUP_MERGE=0 keeps one job per class.
  struct Job { size_t ci; long long j; double weight; int set; };      // j < 0: the whole batch class; set >= 0: merged set `set` (ci unused)
  std::vector<Job> jobs;
  std::vector<size_t> msets[2];
  bool merge = !g->aa_on;
  if (const char* e = getenv("COSMO_HIP_GROUP_MERGE")) merge = merge && atoi(e) != 0;
  for (size_t ci = 0; ci < nc; ++ci) {
    GClass& C = g->cls[ci];
    res[ci].resize(C.members.size());
    if (merge && C.b && C.members.size() == 1 && batch_multi_supported(C.b)) msets[batch_multi_needs_ext(C.b) ? 1 : 0].push_back(ci);
  }
  // a set pays from about a dozen classes on (the streaming form is the slowest kernel per problem; what it saves is host loops and single-workgroup
  // launches): smaller lists keep one job per class and with it the kernel form -- and the bits -- of a uniform batch of each structure
  size_t min_set = 16;
  if (const char* e = getenv("COSMO_HIP_GROUP_MERGE_MIN")) { const int v = atoi(e); if (v >= 2) min_set = (size_t)v; }
  for (int t = 0; t < 2; ++t) if (msets[t].size() < min_set) msets[t].clear();
  std::vector<char> in_set(nc, 0);
  for (int t = 0; t < 2; ++t) { double wsum = 0.0; for (size_t ci : msets[t]) { in_set[ci] = 1; wsum += (double)(g->cls[ci].n + g->cls[ci].m); } if (!msets[t].empty()) jobs.push_back({0, -1, 4.0 * wsum, t}); }
  g->last_merged = (long long)(msets[0].size() + msets[1].size());
  for (size_t ci = 0; ci < nc; ++ci) {
    GClass& C = g->cls[ci];
    if (in_set[ci]) continue;
    const double wt = (double)(C.n + C.m);
    if (C.b) jobs.push_back({ci, -1, wt * (double)C.members.size(), -1});
    else for (size_t j = 0; j < C.members.size(); ++j) jobs.push_back({ci, (long long)j, 64.0 * wt, -1});      // (the launch-per-kernel loop of a single handle is the slow path)
  }
  std::stable_sort(jobs.begin(), jobs.end(), [](const Job& a, const Job& b) { return a.weight > b.weight; });
  std::vector<std::atomic<int32_t>> jrc(nc);
  for (auto& v : jrc) v.store(COSMO_HIP_OK);
  auto run = [&](const Job& jb) {
    if (jb.set >= 0) {
      const std::vector<size_t>& ms = msets[jb.set];
      std::vector<cosmo_hip_batch*> bs; std::vector<cosmo_hip_result> rr(ms.size());
      for (size_t ci : ms) bs.push_back(g->cls[ci].b);
      const int32_t rc = batch_multi_optimize(bs.data(), (int)bs.size(), jb.set == 1, rr.data());
      for (size_t i = 0; i < ms.size(); ++i) { res[ms[i]][0] = rr[i]; if (rc) { int32_t expect = COSMO_HIP_OK; jrc[ms[i]].compare_exchange_strong(expect, rc); } }
      if (rc && !ms.empty() && ms[0] != 0) { /* the error text lives on the first batch of the set */ }
      return;
    }
    GClass& C = g->cls[jb.ci];
    int32_t rc;
    if (jb.j < 0) rc = cosmo_hip_batch_optimize(C.b, res[jb.ci].data());     // (sets the device for its thread; synchronises its own stream only)
    else rc = cosmo_hip_optimize(C.hs[(size_t)jb.j], &res[jb.ci][(size_t)jb.j]);
    if (rc) { int32_t expect = COSMO_HIP_OK; jrc[jb.ci].compare_exchange_strong(expect, rc); }
  };
  int nworkers = 32;
  if (const char* e = getenv("COSMO_HIP_GROUP_WORKERS")) { const int v = atoi(e); if (v >= 1 && v <= 1024) nworkers = v; }
  nworkers = (int)std::min<size_t>((size_t)nworkers, jobs.size());
  g->last_workers = nworkers; g->last_jobs = (long long)jobs.size();
  if (nworkers <= 1) { for (const Job& jb : jobs) run(jb); }
  else {
    std::atomic<size_t> next(0);
    std::vector<std::thread> th;
    for (int t = 0; t < nworkers; ++t) th.emplace_back([&]() { for (;;) { const size_t i = next.fetch_add(1); if (i >= jobs.size()) return; run(jobs[i]); } });
    for (auto& t : th) t.join();
  }
  for (size_t ci = 0; ci < nc; ++ci) rcs[ci] = jrc[ci].load();
  for (size_t ci = 0; ci < nc; ++ci) {
    if (rcs[ci]) {
      std::string detail = g->cls[ci].b ? cosmo_hip_batch_last_error(g->cls[ci].b) : "";
      if (!g->cls[ci].b) for (auto* h : g->cls[ci].hs) { const char* e = cosmo_hip_last_error(h); if (e && *e) { detail = e; break; } }
      return gfail(g, rcs[ci], std::string("optimize of the class of problem ") + std::to_string(g->cls[ci].members[0]) + ": " + detail);
    }
    for (size_t j = 0; j < g->cls[ci].members.size(); ++j) results[g->cls[ci].members[j]] = res[ci][j];
    g->cls[ci].iterates_dirty = false; g->cls[ci].ran = true;
  }
  return COSMO_HIP_OK;
}

extern "C" int32_t cosmo_hip_batch_group_get_iterates(cosmo_hip_batch_group* g, int64_t k, real* w, real* w_prev, real* s, real* mu) {
  GCHECK(g, k);
  if (!g->finalized) return gfail(g, COSMO_HIP_ERR_INVALID, "batch_group_get_iterates: set_params first");
  const GProblem& p = g->prob[(size_t)k];
  GClass& C = g->cls[(size_t)p.cls];
  if (!C.b) { const int32_t rc = cosmo_hip_get_iterates(C.hs[(size_t)p.pos], w, w_prev, s, mu); return rc ? gfail(g, rc, cosmo_hip_last_error(C.hs[(size_t)p.pos])) : COSMO_HIP_OK; }
  const int32_t rc = cosmo_hip_batch_get_iterates(C.b, p.pos, w, w_prev, s, mu);
  if (rc) return gfail(g, rc, cosmo_hip_batch_last_error(C.b));
  return COSMO_HIP_OK;
}

// per problem {ADMM iterations, KKT solves, Krylov iterations in total}: out[3 * nprob]
extern "C" int32_t cosmo_hip_batch_group_get_counters(cosmo_hip_batch_group* g, int64_t* out) {
  if (!g || !g->finalized || !out) return gfail(g, COSMO_HIP_ERR_INVALID, "batch_group_get_counters: set_params first");
  for (auto& C : g->cls) {
    std::vector<int64_t> c(3 * C.members.size());
    if (!C.b) {
      for (size_t j = 0; j < C.members.size(); ++j) {
        int64_t st[8]; const int32_t rc = cosmo_hip_get_stats(C.hs[j], st);
        if (rc) return gfail(g, rc, cosmo_hip_last_error(C.hs[j]));
        c[3 * j] = st[0]; c[3 * j + 1] = st[1]; c[3 * j + 2] = st[2];
      }
      for (size_t j = 0; j < C.members.size(); ++j) for (int t = 0; t < 3; ++t) out[3 * (size_t)C.members[j] + t] = c[3 * j + t];
      continue;
    }
    const int32_t rc = cosmo_hip_batch_get_counters(C.b, c.data());
    if (rc) return gfail(g, rc, cosmo_hip_batch_last_error(C.b));
    for (size_t j = 0; j < C.members.size(); ++j) for (int t = 0; t < 3; ++t) out[3 * (size_t)C.members[j] + t] = c[3 * j + t];
  }
  return COSMO_HIP_OK;
}

// per problem {accelerated steps, safeguard accepted, safeguard declined, memory restarts, active, safeguarding_iter}: out[6 * nprob]
extern "C" int32_t cosmo_hip_batch_group_get_accel_stats(cosmo_hip_batch_group* g, int64_t* out) {
  if (!g || !g->finalized || !out) return gfail(g, COSMO_HIP_ERR_INVALID, "batch_group_get_accel_stats: set_params first");
  for (auto& C : g->cls) {
    std::vector<int64_t> c(6 * C.members.size(), 0);
    if (g->aa_on && !C.b) { for (size_t j = 0; j < C.members.size(); ++j) { const int32_t rc = cosmo_hip_get_accel_stats(C.hs[j], c.data() + 6 * j); if (rc) return gfail(g, rc, cosmo_hip_last_error(C.hs[j])); } }
    else if (g->aa_on) { const int32_t rc = cosmo_hip_batch_get_accel_stats(C.b, c.data()); if (rc) return gfail(g, rc, cosmo_hip_batch_last_error(C.b)); }
    for (size_t j = 0; j < C.members.size(); ++j) for (int t = 0; t < 6; ++t) out[6 * (size_t)C.members[j] + t] = c[6 * j + t];
  }
  return COSMO_HIP_OK;
}
