// chordal_api.cpp -- decomposition driver, compact clique-tree transformation, reverse step and the C ABI
// (restates src/chordal_decomposition/chordal_decomposition.jl and transformations.jl of COSMO.jl v0.8.11).
#include "chordal.hpp"
#include "../../include/cosmo_chordal.h"
#include "../../include/cosmo_hip.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <numeric>
#include <stdexcept>
#include <tuple>

using namespace chordal;

namespace {

thread_local std::string g_err;

inline long long svec_ind(long long i, long long j) {   // mat_to_svec_ind, 1-based (transformations.jl:594-600)
  if (i > j) std::swap(i, j);
  return (j - 1) * j / 2 + i;
}
inline void svec_to_mat(long long k, long long& i, long long& j) {   // svec_to_mat (trees.jl:711-716), 1-based
  long long c = (long long)std::floor((std::sqrt(8.0 * (double)k + 1.0) - 1.0) / 2.0);
  while (c * (c + 1) / 2 < k) ++c;
  while (c > 0 && (c - 1) * c / 2 >= k) --c;
  j = c;
  i = k - (c - 1) * c / 2;
}

struct SparsityPattern {   // src/types.jl:183-215
  SuperNodeTree sntree;
  std::vector<int> ordering;   // 0-based: position in the reordered graph -> row/column of the original matrix
  long long row_start = 0;     // first row (0-based) of the cone in the original problem
  int cone_ind = 0;            // index of the original cone
  int N = 0;                   // side of the matrix
  int type = COSMO_HIP_PSD_TRIANGLE;   // PsdConeTriangle or PsdCone (square; traditional transformation only)
};

struct NewCone { int type; long long dim; int orig; int sp; int clique; };   // clique: post-order index (0-based) or -1

}  // namespace

struct cosmo_chordal {
  long long n = 0, m = 0;
  std::vector<int32_t> type;
  std::vector<long long> dim, off;
  std::vector<SparsityPattern> sp_arr;
  std::vector<int> sp_of_cone;         // original cone -> index in sp_arr or -1
  // decomposed problem
  long long n_new = 0, m_new = 0, num_overlaps = 0;
  std::vector<long long> colptr, rowval;   // 0-based internally
  std::vector<double> nzval, b_new;
  std::vector<NewCone> cones_new;
  bool compact = true;
  // traditional transformation: stacked entry e (column mO.. of the augmented A) -> original row H_I[e] (0-based)
  std::vector<long long> H_I;
};

namespace {

// find_aggregate_sparsity (chordal_decomposition.jl:104-121): svec positions (1-based, ascending) that are nonzero in some
// column of A or in b, plus the diagonal.  `act` = rows of the whole problem that hold a stored entry of A or a nonzero of b
// (computed once for all cones; the reference rescans A.rowval for every cone).
std::vector<long long> aggregate_sparsity(const std::vector<char>& act, long long row0, long long dimc, int N) {
  std::vector<char> a(act.begin() + row0, act.begin() + row0 + dimc);
  for (long long i = 1; i <= N; ++i) a[(size_t)(i * (i + 1) / 2 - 1)] = 1;
  std::vector<long long> out;
  for (long long r = 0; r < dimc; ++r) if (a[(size_t)r]) out.push_back(r + 1);
  return out;
}

// dense helpers for the PSD completion -----------------------------------------------------------------------------------
struct Dense { int r = 0, c = 0; std::vector<double> a; double& at(int i, int j) { return a[(size_t)j * r + i]; } double at(int i, int j) const { return a[(size_t)j * r + i]; } };
Dense zeros(int r, int c) { Dense d; d.r = r; d.c = c; d.a.assign((size_t)r * c, 0.0); return d; }

// X = A \ B by LU with partial pivoting; false when A is numerically singular
bool lu_solve(Dense A, Dense B, Dense& X) {
  const int n = A.r;
  double amax = 0.0;
  for (double v : A.a) amax = std::max(amax, std::fabs(v));
  for (int k = 0; k < n; ++k) {
    int piv = k;
    for (int i = k + 1; i < n; ++i) if (std::fabs(A.at(i, k)) > std::fabs(A.at(piv, k))) piv = i;
    if (!(std::fabs(A.at(piv, k)) > 1e-14 * amax) || amax == 0.0) return false;
    if (piv != k) {
      for (int j = 0; j < n; ++j) std::swap(A.at(k, j), A.at(piv, j));
      for (int j = 0; j < B.c; ++j) std::swap(B.at(k, j), B.at(piv, j));
    }
    for (int i = k + 1; i < n; ++i) {
      const double f = A.at(i, k) / A.at(k, k);
      if (f == 0.0) continue;
      for (int j = k; j < n; ++j) A.at(i, j) -= f * A.at(k, j);
      for (int j = 0; j < B.c; ++j) B.at(i, j) -= f * B.at(k, j);
    }
  }
  X = zeros(n, B.c);
  for (int j = 0; j < B.c; ++j)
    for (int i = n - 1; i >= 0; --i) {
      double s = B.at(i, j);
      for (int k = i + 1; k < n; ++k) s -= A.at(i, k) * X.at(k, j);
      X.at(i, j) = s / A.at(i, i);
    }
  return true;
}
// X = pinv(A) B for symmetric A (cyclic Jacobi eigendecomposition) -- the reference's fallback `pinv(Waa) * Wαν`
void pinv_solve_sym(Dense A, const Dense& B, Dense& X) {
  const int n = A.r;
  Dense V = zeros(n, n);
  for (int i = 0; i < n; ++i) V.at(i, i) = 1.0;
  for (int sweep = 0; sweep < 60; ++sweep) {
    double offd = 0.0;
    for (int p = 0; p < n; ++p) for (int q = p + 1; q < n; ++q) offd += A.at(p, q) * A.at(p, q);
    if (offd < 1e-300) break;
    for (int p = 0; p < n; ++p)
      for (int q = p + 1; q < n; ++q) {
        if (std::fabs(A.at(p, q)) < 1e-300) continue;
        const double th = (A.at(q, q) - A.at(p, p)) / (2.0 * A.at(p, q));
        const double t = (th >= 0 ? 1.0 : -1.0) / (std::fabs(th) + std::sqrt(th * th + 1.0));
        const double c = 1.0 / std::sqrt(t * t + 1.0), s = t * c;
        for (int k = 0; k < n; ++k) { const double akp = A.at(k, p), akq = A.at(k, q); A.at(k, p) = c * akp - s * akq; A.at(k, q) = s * akp + c * akq; }
        for (int k = 0; k < n; ++k) { const double apk = A.at(p, k), aqk = A.at(q, k); A.at(p, k) = c * apk - s * aqk; A.at(q, k) = s * apk + c * aqk; }
        for (int k = 0; k < n; ++k) { const double vkp = V.at(k, p), vkq = V.at(k, q); V.at(k, p) = c * vkp - s * vkq; V.at(k, q) = s * vkp + c * vkq; }
      }
  }
  double lmax = 0.0;
  for (int i = 0; i < n; ++i) lmax = std::max(lmax, std::fabs(A.at(i, i)));
  const double tol = lmax * n * 2.220446049250313e-16;
  X = zeros(n, B.c);
  for (int k = 0; k < n; ++k) {
    const double l = A.at(k, k);
    if (!(std::fabs(l) > tol)) continue;
    for (int j = 0; j < B.c; ++j) {
      double vb = 0.0;
      for (int i = 0; i < n; ++i) vb += V.at(i, k) * B.at(i, j);
      vb /= l;
      for (int i = 0; i < n; ++i) X.at(i, j) += V.at(i, k) * vb;
    }
  }
}

// psd_complete! (chordal_decomposition.jl:259-311; Vandenberghe & Andersen, Chordal graphs and semidefinite optimization, p. 362)
void psd_complete(Dense& A, int N, const SuperNodeTree& t, const std::vector<int>& p) {
  Dense W = zeros(N, N);
  for (int i = 0; i < N; ++i)
    for (int j = 0; j < N; ++j) { const int a = p[i], b = p[j]; W.at(i, j) = (a <= b) ? A.at(a, b) : A.at(b, a); }   // Symmetric(A, :U)[p, p]
  for (int jj = t.num - 2; jj >= 0; --jj) {
    const int c = t.snd_post[jj];
    std::vector<int> nu(t.snd[c].begin(), t.snd[c].end()), al(t.sep[c].begin(), t.sep[c].end()), eta;
    const int i0 = nu.front();
    for (int x = i0 + 1; x < N; ++x) if (!t.snd[c].count(x) && !t.sep[c].count(x)) eta.push_back(x);
    if (al.empty() || eta.empty()) continue;   // nothing to complete through an empty separator: the block stays zero
    Dense Waa = zeros((int)al.size(), (int)al.size()), Wan = zeros((int)al.size(), (int)nu.size());
    for (size_t a = 0; a < al.size(); ++a) {
      for (size_t b = 0; b < al.size(); ++b) Waa.at((int)a, (int)b) = W.at(al[a], al[b]);
      for (size_t b = 0; b < nu.size(); ++b) Wan.at((int)a, (int)b) = W.at(al[a], nu[b]);
    }
    Dense Y;
    if (!lu_solve(Waa, Wan, Y)) pinv_solve_sym(Waa, Wan, Y);
    for (size_t e = 0; e < eta.size(); ++e)
      for (size_t b = 0; b < nu.size(); ++b) {
        double s = 0.0;
        for (size_t a = 0; a < al.size(); ++a) s += W.at(eta[e], al[a]) * Y.at((int)a, (int)b);
        W.at(eta[e], nu[b]) = s;
        W.at(nu[b], eta[e]) = s;
      }
  }
  std::vector<int> ip(N);
  for (int i = 0; i < N; ++i) ip[p[i]] = i;
  for (int i = 0; i < N; ++i) for (int j = 0; j < N; ++j) A.at(i, j) = W.at(ip[i], ip[j]);
}

// get_block_indices (transformations.jl:401-431): (i, j, flag) of a clique block in svec order of the sorted clique; flag 0 = overlap
std::vector<std::tuple<int, int, int>> block_indices(const std::vector<int>& snd, const std::vector<int>& sep, int Nv) {
  std::vector<std::tuple<int, int, int>> out;
  for (int j : sep) for (int i : sep) if (i <= j) out.emplace_back(i, j, 0);
  for (int j : snd) for (int i : snd) if (i <= j) out.emplace_back(i, j, 1);
  for (int i : snd) for (int j : sep) out.emplace_back(std::min(i, j), std::max(i, j), 1);
  std::stable_sort(out.begin(), out.end(), [&](const auto& a, const auto& b) {
    return (long long)std::get<1>(a) * Nv + std::get<0>(a) < (long long)std::get<1>(b) * Nv + std::get<0>(b);
  });
  return out;
}

void run_decompose(cosmo_chordal& C, const int64_t* Ap, const int64_t* Ai, const double* Ax, const double* b, const cosmo_chordal_options& opt) {
  const size_t nc = C.type.size();
  C.sp_of_cone.assign(nc, -1);
  const int64_t* ord_in = opt.orderings;
  std::vector<char> act((size_t)C.m, 0);
  { const long long nnz0 = Ap[C.n] - 1;
    for (long long k = 0; k < nnz0; ++k) act[(size_t)(Ai[k] - 1)] = 1;
    for (long long r = 0; r < C.m; ++r) if (b[r] != 0.0) act[(size_t)r] = 1; }
  // ---- find_sparsity_patterns! (chordal_decomposition.jl:41-77) ----
  for (size_t k = 0; k < nc; ++k) {
    const bool tri = C.type[k] == COSMO_HIP_PSD_TRIANGLE, sq = C.type[k] == COSMO_HIP_PSD_SQUARE && !C.compact;
    if (!tri && !sq) continue;
    const long long dimc = C.dim[k];
    const int N = tri ? (int)((std::llround(std::floor(std::sqrt(1.0 + 8.0 * (double)dimc))) - 1) / 2) : (int)std::llround(std::sqrt((double)dimc));
    if (tri && (long long)N * (N + 1) / 2 != dimc) throw std::runtime_error("PsdConeTriangle dimension is not triangular");
    if (sq && (long long)N * N != dimc) throw std::runtime_error("PsdCone dimension is not a square");
    const int64_t* my_ord = ord_in;
    if (ord_in) ord_in += N;
    std::vector<long long> csp;
    if (tri) csp = aggregate_sparsity(act, C.off[k], dimc, N);
    else {
      // square cone: the reference flags the positions vec_dim(i, C) = i^2 instead of the diagonal (chordal_decomposition.jl:109-111
      // with trees.jl:261) -- kept: it only adds the edges those positions stand for
      std::vector<char> a(act.begin() + C.off[k], act.begin() + C.off[k] + dimc);
      for (long long i = 1; i <= N; ++i) a[(size_t)(i * i - 1)] = 1;
      for (long long r = 0; r < dimc; ++r) if (a[(size_t)r]) csp.push_back(r + 1);
    }
    if ((long long)csp.size() >= dimc) continue;               // dense cone: DenseEquivalent (:56-61)
    // find_graph! (trees.jl:634-645)
    std::vector<IntSet> adj(N);
    for (long long r : csp) {
      long long i, j;
      if (tri) svec_to_mat(r, i, j);
      else { i = (r - 1) % N + 1; j = (r - 1) / N + 1; }           // row_ind_to_matrix_indices (trees.jl:659-675)
      if (i != j) { adj[i - 1].insert((int)j - 1); adj[j - 1].insert((int)i - 1); }
    }
    std::vector<int> perm(N);
    if (my_ord) {
      std::vector<char> seen(N, 0);
      for (int i = 0; i < N; ++i) {
        const long long v = my_ord[i] - 1;
        if (v < 0 || v >= N || seen[v]) throw std::runtime_error("orderings: not a permutation");
        seen[v] = 1; perm[i] = (int)v;
      }
    } else {
      perm = minimum_degree_ordering(N, adj);
    }
    LPattern L = symbolic_ldl(N, adj, perm);
    connect_graph(L);
    SparsityPattern sp;
    sp.N = N; sp.row_start = C.off[k]; sp.cone_ind = (int)k; sp.ordering = perm; sp.type = C.type[k];
    build_supernode_tree(sp.sntree, L, opt.merge_strategy, opt.t_fill, opt.t_size);
    if (sp.sntree.num > 1) merge_cliques(sp.sntree);            // SparsityPattern constructor (types.jl:192-215)
    reorder_snd_consecutively(sp.sntree, sp.ordering);
    calculate_block_dimensions(sp.sntree);
    if (sp.sntree.num == 1) continue;                          // one clique left: do not decompose (:68-71)
    C.sp_of_cone[k] = (int)C.sp_arr.size();
    C.sp_arr.push_back(std::move(sp));
  }
  if (!C.compact) {
    // ---- find_decomposition_matrix! + augment_system! (transformations.jl:4-138): s = H sbar, [A H; 0 -I] [x; sbar] + [s; -sbar'] ... ----
    C.cones_new.clear(); C.H_I.clear();
    C.cones_new.push_back({COSMO_HIP_ZERO, C.m, -1, -1, -1});
    for (size_t k = 0; k < nc; ++k) {
      const int spi = C.sp_of_cone[k];
      if (spi < 0) {
        for (long long r = 0; r < C.dim[k]; ++r) C.H_I.push_back(C.off[k] + r);
        C.cones_new.push_back({C.type[k], C.dim[k], (int)k, -1, -1});
        continue;
      }
      const SparsityPattern& sp = C.sp_arr[spi];
      for (int iii = 0; iii < sp.sntree.num; ++iii) {             // ascending post order (transformations.jl:71-86)
        std::vector<int> cl = get_clique(sp.sntree, iii), c;
        for (int v : cl) c.push_back(sp.ordering[v] + 1);
        std::sort(c.begin(), c.end());
        long long rows = 0;
        for (int vj : c) for (int vi : c) {
          if (sp.type == COSMO_HIP_PSD_TRIANGLE) { if (vi <= vj) { C.H_I.push_back(sp.row_start + svec_ind(vi, vj) - 1); ++rows; } }
          else { C.H_I.push_back(sp.row_start + (long long)(vj - 1) * sp.N + vi - 1); ++rows; }
        }
        C.cones_new.push_back({sp.type, rows, (int)k, spi, iii});
      }
    }
    const long long nH = (long long)C.H_I.size();
    C.n_new = C.n + nH; C.m_new = C.m + nH; C.num_overlaps = nH;
    C.colptr.assign((size_t)C.n_new + 1, 0); C.rowval.clear(); C.nzval.clear();
    for (long long col = 0; col < C.n; ++col) {
      for (long long kk = Ap[col] - 1; kk < Ap[col + 1] - 1; ++kk) { C.rowval.push_back(Ai[kk] - 1); C.nzval.push_back(Ax[kk]); }
      C.colptr[(size_t)col + 1] = (long long)C.rowval.size();
    }
    for (long long e = 0; e < nH; ++e) {                           // column n + e: H entry (row H_I[e], +1) then the -I entry (row m + e)
      C.rowval.push_back(C.H_I[(size_t)e]); C.nzval.push_back(1.0);
      C.rowval.push_back(C.m + e); C.nzval.push_back(-1.0);
      C.colptr[(size_t)(C.n + e) + 1] = (long long)C.rowval.size();
    }
    C.b_new.assign((size_t)C.m_new, 0.0);
    for (long long r = 0; r < C.m; ++r) C.b_new[(size_t)r] = b[r];
    return;
  }
  // ---- augment_clique_based! (transformations.jl:152-200) ----
  const long long nnzA = Ap[C.n] - 1;
  std::vector<long long> newrow((size_t)C.m, -1);               // original row -> row of the decomposed problem (non-overlap entries)
  std::vector<std::tuple<long long, long long, double>> extra;  // (row, col, val) of the overlap columns
  long long row_ptr = 0, ovl_col = C.n;
  C.cones_new.clear();
  for (size_t k = 0; k < nc; ++k) {
    const int spi = C.sp_of_cone[k];
    if (spi < 0) {                                              // pass-through cone (transformations.jl:236-273)
      for (long long r = 0; r < C.dim[k]; ++r) newrow[(size_t)(C.off[k] + r)] = row_ptr + r;
      C.cones_new.push_back({C.type[k], C.dim[k], (int)k, -1, -1});
      row_ptr += C.dim[k];
      continue;
    }
    const SparsityPattern& sp = C.sp_arr[spi];
    const SuperNodeTree& t = sp.sntree;
    const int Nc = t.num;
    // clique_rows_map (transformations.jl:441-452): rows of every clique block, keyed by clique index
    std::map<int, long long> clique_row_start;
    { long long rs = row_ptr; for (int i = Nc - 1; i >= 0; --i) { clique_row_start[t.snd_post[i]] = rs; rs += (long long)t.nBlk[i] * (t.nBlk[i] + 1) / 2; } }
    for (int iii = Nc - 1; iii >= 0; --iii) {                   // descending topological order (transformations.jl:289-325)
      const int c = t.snd_post[iii];
      std::vector<int> sep, snd;
      for (int v : t.sep[c]) sep.push_back(sp.ordering[v] + 1);
      for (int v : t.snd[c]) snd.push_back(sp.ordering[v] + 1);
      const auto blocks = block_indices(snd, sep, sp.N);
      std::vector<int> par_clique;
      long long par_row_start = 0;
      if (iii != Nc - 1) {
        const int pc = t.snd_par[c];
        par_row_start = clique_row_start.at(pc);
        for (int v : t.snd[pc]) par_clique.push_back(sp.ordering[v] + 1);
        for (int v : t.sep[pc]) par_clique.push_back(sp.ordering[v] + 1);
        std::sort(par_clique.begin(), par_clique.end());
      }
      long long counter = 0;
      for (const auto& blk : blocks) {
        const long long new_row = row_ptr + counter;
        const int i = std::get<0>(blk), j = std::get<1>(blk);
        if (std::get<2>(blk) == 0) {                            // overlap with the parent clique: +1 here, -1 in the parent's row
          const long long ir = std::lower_bound(par_clique.begin(), par_clique.end(), i) - par_clique.begin() + 1;
          const long long jr = std::lower_bound(par_clique.begin(), par_clique.end(), j) - par_clique.begin() + 1;
          extra.emplace_back(new_row, ovl_col, 1.0);
          extra.emplace_back(par_row_start + svec_ind(ir, jr) - 1, ovl_col, -1.0);
          ++ovl_col;
        } else {
          newrow[(size_t)(sp.row_start + svec_ind(i, j) - 1)] = new_row;
        }
        ++counter;
      }
      const long long num_rows = (long long)t.nBlk[iii] * (t.nBlk[iii] + 1) / 2;
      C.cones_new.push_back({COSMO_HIP_PSD_TRIANGLE, num_rows, (int)k, spi, iii});
      row_ptr += num_rows;
    }
  }
  C.m_new = row_ptr;
  C.num_overlaps = ovl_col - C.n;
  C.n_new = ovl_col;
  // assemble the new A (CSC, rows sorted within columns) and b
  std::vector<std::vector<std::pair<long long, double>>> cols((size_t)C.n_new);
  for (long long col = 0; col < C.n; ++col)
    for (long long k = Ap[col] - 1; k < Ap[col + 1] - 1; ++k) {
      const long long nr = newrow[(size_t)(Ai[k] - 1)];
      if (nr < 0) {
        if (Ax[k] != 0.0) throw std::runtime_error("internal: a nonzero of A lies outside every clique");
        continue;                                               // explicitly stored zero outside the sparsity pattern
      }
      cols[(size_t)col].emplace_back(nr, Ax[k]);
    }
  for (auto& e : extra) cols[(size_t)std::get<1>(e)].emplace_back(std::get<0>(e), std::get<2>(e));
  (void)nnzA;
  C.colptr.assign((size_t)C.n_new + 1, 0);
  C.rowval.clear(); C.nzval.clear();
  for (long long col = 0; col < C.n_new; ++col) {
    auto& v = cols[(size_t)col];
    std::stable_sort(v.begin(), v.end(), [](const auto& a, const auto& b) { return a.first < b.first; });
    for (auto& e : v) { C.rowval.push_back(e.first); C.nzval.push_back(e.second); }
    C.colptr[(size_t)col + 1] = (long long)C.rowval.size();
  }
  C.b_new.assign((size_t)C.m_new, 0.0);
  for (long long r = 0; r < C.m; ++r) if (b[r] != 0.0) {
    if (newrow[(size_t)r] < 0) throw std::runtime_error("internal: a nonzero of b lies outside every clique");
    C.b_new[(size_t)newrow[(size_t)r]] = b[r];
  }
}

int fail(const std::exception& e) { g_err = e.what(); return 1; }

void sets_from_csr(int64_t count, const int64_t* ptr, const int64_t* idx, std::vector<IntSet>& out) {
  out.assign((size_t)count, IntSet());
  for (int64_t k = 0; k < count; ++k) for (int64_t p = ptr[k]; p < ptr[k + 1]; ++p) out[(size_t)k].insert((int)idx[p] - 1);
}

}  // namespace

extern "C" {

const char* cosmo_chordal_last_error(void) { return g_err.c_str(); }

void cosmo_chordal_default_options(cosmo_chordal_options* o) {
  if (!o) return;
  o->merge_strategy = COSMO_CHORDAL_CLIQUE_GRAPH_MERGE; o->t_fill = 8; o->t_size = 8; o->orderings = nullptr; o->compact_transformation = 1;
}

int32_t cosmo_chordal_decompose(int64_t n, int64_t m, const int64_t* A_colptr, const int64_t* A_rowval, const double* A_nzval, const double* b,
                                int64_t ncones, const int32_t* type, const int64_t* dim, const cosmo_chordal_options* opt, cosmo_chordal** out) {
  if (!out) return 1;
  *out = nullptr;
  try {
    if (n < 0 || m < 0 || !A_colptr || (m > 0 && !b) || ncones < 0) throw std::runtime_error("bad arguments");
    cosmo_chordal_options o;
    cosmo_chordal_default_options(&o);
    if (opt) o = *opt;
    if (o.merge_strategy < 0 || o.merge_strategy > 2) throw std::runtime_error("unknown merge strategy");
    auto* C = new cosmo_chordal();
    C->n = n; C->m = m; C->compact = o.compact_transformation != 0;
    long long off = 0;
    for (int64_t k = 0; k < ncones; ++k) { C->type.push_back(type[k]); C->dim.push_back(dim[k]); C->off.push_back(off); off += dim[k]; }
    if (off != m) { delete C; throw std::runtime_error("cone dimensions do not sum to m"); }
    try { run_decompose(*C, A_colptr, A_rowval, A_nzval, b, o); } catch (...) { delete C; throw; }
    *out = C;
    return 0;
  } catch (const std::exception& e) { return fail(e); }
}

void cosmo_chordal_free(cosmo_chordal* c) { delete c; }

int32_t cosmo_chordal_sizes(const cosmo_chordal* c, int64_t s[6]) {
  if (!c || !s) return 1;
  s[0] = c->n_new; s[1] = c->m_new; s[2] = (int64_t)c->rowval.size(); s[3] = (int64_t)c->cones_new.size(); s[4] = (int64_t)c->sp_arr.size(); s[5] = c->num_overlaps;
  return 0;
}

int32_t cosmo_chordal_get_problem(const cosmo_chordal* c, int64_t* colptr, int64_t* rowval, double* nzval, double* b, int32_t* type, int64_t* dim,
                                  int64_t* cone_map, int64_t* clique_of) {
  if (!c) return 1;
  if (colptr) for (size_t i = 0; i < c->colptr.size(); ++i) colptr[i] = c->colptr[i] + 1;
  if (rowval) for (size_t i = 0; i < c->rowval.size(); ++i) rowval[i] = c->rowval[i] + 1;
  if (nzval) std::copy(c->nzval.begin(), c->nzval.end(), nzval);
  if (b) std::copy(c->b_new.begin(), c->b_new.end(), b);
  for (size_t k = 0; k < c->cones_new.size(); ++k) {
    if (type) type[k] = c->cones_new[k].type;
    if (dim) dim[k] = c->cones_new[k].dim;
    if (cone_map) cone_map[k] = c->cones_new[k].orig + 1;
    if (clique_of) clique_of[k] = c->cones_new[k].clique + 1;
  }
  return 0;
}

int32_t cosmo_chordal_num_cliques(const cosmo_chordal* c, int64_t cone, int64_t* num, int64_t* total) {
  if (!c || cone < 1 || cone > (int64_t)c->sp_of_cone.size()) return 1;
  const int spi = c->sp_of_cone[(size_t)cone - 1];
  *num = 0; *total = 0;
  if (spi < 0) return 0;
  const SuperNodeTree& t = c->sp_arr[spi].sntree;
  *num = t.num;
  for (int i = 0; i < t.num; ++i) *total += t.nBlk[i];
  return 0;
}

int32_t cosmo_chordal_get_cliques(const cosmo_chordal* c, int64_t cone, int64_t* ptr, int64_t* vertices) {
  if (!c || cone < 1 || cone > (int64_t)c->sp_of_cone.size()) return 1;
  const int spi = c->sp_of_cone[(size_t)cone - 1];
  if (spi < 0) return 0;
  const SparsityPattern& sp = c->sp_arr[spi];
  int64_t p = 0;
  for (int i = 0; i < sp.sntree.num; ++i) {
    ptr[i] = p;
    std::vector<int> cl = get_clique(sp.sntree, i);
    std::vector<int64_t> orig;
    for (int v : cl) orig.push_back(sp.ordering[v] + 1);
    std::sort(orig.begin(), orig.end());
    for (int64_t v : orig) vertices[p++] = v;
  }
  ptr[sp.sntree.num] = p;
  return 0;
}

int32_t cosmo_chordal_merge_log(const cosmo_chordal* c, int64_t cone, int64_t* nd, int64_t* nm, int64_t* pairs, int32_t* decisions, int64_t cap) {
  if (!c || cone < 1 || cone > (int64_t)c->sp_of_cone.size()) return 1;
  const int spi = c->sp_of_cone[(size_t)cone - 1];
  *nd = 0; *nm = 0;
  if (spi < 0) return 0;
  const MergeLog& L = c->sp_arr[spi].sntree.merge_log;
  *nd = (int64_t)L.decisions.size(); *nm = L.num;
  for (int64_t i = 0; i < *nd && i < cap; ++i) { pairs[2 * i] = L.clique_pairs[(size_t)i][0] + 1; pairs[2 * i + 1] = L.clique_pairs[(size_t)i][1] + 1; decisions[i] = L.decisions[(size_t)i]; }
  return 0;
}

// reverse_decomposition! with the compact transformation (chordal_decomposition.jl:126-215)
int32_t cosmo_chordal_reverse(const cosmo_chordal* c, const double* s_dec, const double* mu_dec, double* s_out, double* mu_out, int32_t complete_dual) {
  if (!c || !s_dec || !mu_dec || !s_out || !mu_out) return 1;
  try {
    std::fill(s_out, s_out + c->m, 0.0);
    std::fill(mu_out, mu_out + c->m, 0.0);
    long long row_start = 0;
    if (!c->compact) {
      // s = H sbar ; mu = H mubar averaged over the overlapping blocks (chordal_decomposition.jl:143-146, 152-168)
      std::vector<int> cnt((size_t)c->m, 0);
      for (size_t e = 0; e < c->H_I.size(); ++e) {
        const long long r = c->H_I[e];
        s_out[r] += s_dec[c->m + (long long)e];
        mu_out[r] += mu_dec[c->m + (long long)e];
        cnt[(size_t)r] += 1;
      }
      for (long long r = 0; r < c->m; ++r) if (cnt[(size_t)r] > 1) mu_out[r] = mu_out[r] / cnt[(size_t)r];
    }
    for (const NewCone& nc : c->cones_new) {
      if (!c->compact) break;
      const long long o0 = c->off[(size_t)nc.orig];
      if (nc.sp < 0) {                                           // add_blocks! for non-decomposed cones (:183-188)
        for (long long r = 0; r < nc.dim; ++r) { s_out[o0 + r] = s_dec[row_start + r]; mu_out[o0 + r] = mu_dec[row_start + r]; }
        row_start += nc.dim;
        continue;
      }
      const SparsityPattern& sp = c->sp_arr[(size_t)nc.sp];
      std::vector<int> cl = get_clique(sp.sntree, nc.clique);
      std::vector<int> clique;
      for (int v : cl) clique.push_back(sp.ordering[v] + 1);
      std::sort(clique.begin(), clique.end());
      long long counter = 0;
      for (int j : clique) for (int i : clique) if (i <= j) {    // (:190-215)
        const long long offset = svec_ind(i, j) - 1;
        s_out[o0 + offset] += s_dec[row_start + counter];
        mu_out[o0 + offset] = mu_dec[row_start + counter];       // overlapping entries are overwritten
        ++counter;
      }
      row_start += counter;
    }
    if (complete_dual) {                                         // psd_completion! (:220-257): complete y = -mu
      const double isq2 = 1.0 / std::sqrt(2.0), sq2 = std::sqrt(2.0);
      for (const SparsityPattern& sp : c->sp_arr) {
        const int N = sp.N;
        const long long o0 = sp.row_start;
        Dense X = zeros(N, N);
        long long k = 0;
        if (sp.type == COSMO_HIP_PSD_SQUARE) {                     // complete!(mu, ::PsdCone) (:231-243): reshape, upper triangle is used
          for (int j = 0; j < N; ++j) for (int i = 0; i < N; ++i) X.at(i, j) = -mu_out[o0 + (long long)j * N + i];
          psd_complete(X, N, sp.sntree, sp.ordering);
          for (int j = 0; j < N; ++j) for (int i = 0; i < N; ++i) mu_out[o0 + (long long)j * N + i] = -X.at(i, j);
          continue;
        }
        for (int j = 0; j < N; ++j) for (int i = 0; i <= j; ++i) { const double v = -mu_out[o0 + k++]; X.at(i, j) = (i == j) ? v : isq2 * v; }
        psd_complete(X, N, sp.sntree, sp.ordering);
        k = 0;
        for (int j = 0; j < N; ++j) for (int i = 0; i <= j; ++i) { const double v = (i == j) ? X.at(i, j) : sq2 * X.at(i, j); mu_out[o0 + k++] = -v; }
      }
    }
    return 0;
  } catch (const std::exception& e) { return fail(e); }
}

// ---- golden-test hooks -------------------------------------------------------------------------------------------------
int32_t cosmo_chordal_test_merge_tree(int64_t ncl, const int64_t* snd_ptr, const int64_t* snd, const int64_t* sep_ptr, const int64_t* sep, const int64_t* par,
                                      const int64_t* snd_post, int64_t nvertices, int32_t strategy, int64_t* nd, int64_t* nm, int64_t* pairs,
                                      int32_t* decisions, int64_t* par_out, int64_t cap) {
  try {
    SuperNodeTree t;
    t.strategy = strategy;
    sets_from_csr(ncl, snd_ptr, snd, t.snd);
    sets_from_csr(ncl, sep_ptr, sep, t.sep);
    t.snd_par.resize((size_t)ncl);
    for (int64_t i = 0; i < ncl; ++i) t.snd_par[(size_t)i] = (par[i] == 0) ? -1 : (par[i] < 0 ? -2 : (int)par[i] - 1);
    t.snd_child = child_from_par(t.snd_par);
    t.snd_post.resize((size_t)ncl);
    for (int64_t i = 0; i < ncl; ++i) t.snd_post[(size_t)i] = (int)snd_post[i] - 1;
    t.post.resize((size_t)nvertices);
    std::iota(t.post.begin(), t.post.end(), 0);
    t.num = (int)ncl;
    if (strategy == CLIQUE_GRAPH_MERGE) {                        // the reference's test turns the tree into a graph first (t.snd = union.(snd, sep))
      for (int64_t i = 0; i < ncl; ++i) t.snd[(size_t)i].insert(t.sep[(size_t)i].begin(), t.sep[(size_t)i].end());
      // the reference test keeps snd_par from the example tree when no merge happens; merge_cliques! overwrites it otherwise
    }
    const std::vector<int> par_before = t.snd_par;
    merge_cliques(t);
    *nd = (int64_t)t.merge_log.decisions.size(); *nm = t.merge_log.num;
    for (int64_t i = 0; i < *nd && i < cap; ++i) { pairs[2 * i] = t.merge_log.clique_pairs[(size_t)i][0] + 1; pairs[2 * i + 1] = t.merge_log.clique_pairs[(size_t)i][1] + 1; decisions[i] = t.merge_log.decisions[(size_t)i]; }
    for (int64_t i = 0; i < ncl; ++i) { const int p = t.snd_par[(size_t)i]; par_out[i] = (p == -1) ? 0 : (p == -2 ? -1 : p + 1); }
    (void)par_before;
    return 0;
  } catch (const std::exception& e) { return fail(e); }
}

int32_t cosmo_chordal_test_reduced_clique_graph(int64_t ncl, const int64_t* snd_ptr, const int64_t* snd, int64_t nsep, const int64_t* sep_ptr,
                                                const int64_t* sep, int64_t* nedges, int64_t* rows, int64_t* cols, double* weights, int32_t* permissible,
                                                int64_t cap) {
  try {
    SuperNodeTree t;
    t.strategy = CLIQUE_GRAPH_MERGE;
    sets_from_csr(ncl, snd_ptr, snd, t.snd);
    sets_from_csr(nsep, sep_ptr, sep, t.sep);
    t.num = (int)ncl;
    t.snd_par.assign((size_t)ncl, -2);
    t.snd_child.assign((size_t)ncl, IntSet());
    initialise(t);
    int64_t k = 0;
    for (auto& kv : t.edges.e) {
      if (k < cap) {
        rows[k] = kv.first.second + 1; cols[k] = kv.first.first + 1; weights[k] = kv.second;
        permissible[k] = ispermissible(kv.first.second, kv.first.first, t.adjacency_table, t.snd) ? 1 : 0;
      }
      ++k;
    }
    *nedges = k;
    return 0;
  } catch (const std::exception& e) { return fail(e); }
}

}  // extern "C"
