// chordal_tree.cpp -- elimination tree, supernodes, clique merging (restates src/chordal_decomposition/trees.jl,
// clique_graph.jl, clique_merging.jl of COSMO.jl v0.8.11; 0-based, see chordal.hpp).
#include "chordal.hpp"

#include <algorithm>
#include <cmath>
#include <functional>
#include <numeric>
#include <stdexcept>

namespace chordal {

// ---------------------------------------------------------------------------------------------------------------------
// trees.jl
// ---------------------------------------------------------------------------------------------------------------------
// etree / find_parent_direct (trees.jl:158-167, 560-565): the parent of v is the first sub-diagonal nonzero of column v
std::vector<int> etree(const LPattern& L) {
  std::vector<int> par(L.n, -1);
  for (int v = 0; v + 1 < L.n; ++v) {
    if (L.cols[v].empty()) throw std::runtime_error("etree: unconnected column (connect_graph! must run first)");
    par[v] = L.cols[v].front();
  }
  return par;
}

std::vector<IntSet> child_from_par(const std::vector<int>& par) {   // trees.jl:198-206
  std::vector<IntSet> child(par.size());
  for (int i = 0; i < (int)par.size(); ++i)
    if (par[i] >= 0) child[par[i]].insert(i);
  return child;
}

// post_order (trees.jl:172-195): depth-first search from the root; the vertex popped first gets the highest order.  The
// reference pushes the children in the iteration order of a Julia Set (hash order); here: ascending, so the child with the
// largest index is visited first.  Entries with par == -2 (removed by merges) are never reached and are cut off.
std::vector<int> post_order(const std::vector<int>& par, const std::vector<IntSet>& child, int Nc) {
  const int n = (int)par.size();
  std::vector<int> order(n, Nc + 1);
  int root = -1;
  for (int i = 0; i < n; ++i) if (par[i] == -1) { root = i; break; }
  if (root < 0) throw std::runtime_error("post_order: no root");
  std::vector<int> stack{root};
  int iii = Nc;
  while (!stack.empty()) {
    const int v = stack.back(); stack.pop_back();
    order[v] = iii--;
    for (int c : child[v]) stack.push_back(c);
  }
  std::vector<int> post(n);
  std::iota(post.begin(), post.end(), 0);
  std::stable_sort(post.begin(), post.end(), [&](int a, int b) { return order[a] < order[b]; });
  if (Nc != n) post.resize(Nc);
  return post;
}

std::vector<int> higher_degrees(const LPattern& L) {   // trees.jl:569-579
  std::vector<int> deg(L.n, 0);
  for (int v = 0; v + 1 < L.n; ++v) deg[v] = (int)L.cols[v].size();
  return deg;
}

// Pothen & Sun, Compact clique tree data structures in sparse matrix factorizations (1989); trees.jl:386-457
void pothen_sun(const std::vector<int>& par, const std::vector<int>& post, const std::vector<int>& degrees, std::vector<int>& sn_par,
                std::vector<int>& sn_ind) {
  const int N = (int)par.size();
  sn_ind.assign(N, -1);                 // < 0: representative vertex (the magnitude counts members); >= 0: the representative
  std::vector<int> supernode_par(N, -1);
  std::vector<std::vector<int>> children(N);
  int root_ind = -1;
  for (int i = 0; i < N; ++i) if (par[i] == -1) { root_ind = i; break; }
  for (int v : post) {
    if (par[v] == -1) children[root_ind].push_back(v);
    else children[par[v]].push_back(v);
    if (par[v] != -1) {
      const int pv = par[v];
      if (degrees[v] - 1 == degrees[pv] && sn_ind[pv] == -1) {
        if (sn_ind[v] < 0) { sn_ind[pv] = v; sn_ind[v] -= 1; }                 // case A: v is a representative vertex
        else { sn_ind[pv] = sn_ind[v]; sn_ind[sn_ind[v]] -= 1; }              // case B
      } else {
        if (sn_ind[v] < 0) supernode_par[v] = v;
        else supernode_par[sn_ind[v]] = sn_ind[v];
      }
    }
    const int k = sn_ind[v] < 0 ? v : sn_ind[v];
    for (int w : children[v]) {
      const int l = sn_ind[w] < 0 ? w : sn_ind[w];
      if (l != k) supernode_par[l] = k;
    }
  }
  std::vector<int> reprv;
  for (int v = 0; v < N; ++v) if (sn_ind[v] < 0) reprv.push_back(v);
  sn_par.assign(reprv.size(), -1);
  for (size_t i = 0; i < reprv.size(); ++i) {
    const int rp = supernode_par[reprv[i]];
    auto it = (rp < 0) ? reprv.end() : std::find(reprv.begin(), reprv.end(), rp);
    sn_par[i] = (it == reprv.end()) ? -1 : (int)(it - reprv.begin());
  }
}

// connect_graph! (trees.jl:587-604): a column without sub-diagonal entries starts a new connected component; link it to the
// next vertex so that the elimination tree has one root
void connect_graph(LPattern& L) {
  for (int j = 0; j + 1 < L.n; ++j)
    if (L.cols[j].empty()) L.cols[j].push_back(j + 1);
}

// SuperNodeTree(L, merge_strategy) (trees.jl:74-107)
void build_supernode_tree(SuperNodeTree& t, const LPattern& L, int strategy, int t_fill, int t_size) {
  t = SuperNodeTree();
  t.strategy = strategy; t.t_fill = t_fill; t.t_size = t_size;
  t.par = etree(L);
  const std::vector<IntSet> child = child_from_par(t.par);
  t.post = post_order(t.par, child, (int)t.par.size());
  const std::vector<int> degrees = higher_degrees(L);
  std::vector<int> sn_par, sn_ind;
  pothen_sun(t.par, t.post, degrees, sn_par, sn_ind);
  // find_supernodes (trees.jl:467-486): supernodes in ascending order of their representative vertex
  const int N = (int)t.par.size();
  std::vector<IntSet> snode(N);
  for (int i = 0; i < N; ++i) snode[sn_ind[i] < 0 ? i : sn_ind[i]].insert(i);
  for (auto& s : snode) if (!s.empty()) t.snd.push_back(s);
  t.snd_par = sn_par;
  t.snd_child = child_from_par(t.snd_par);
  t.snd_post = post_order(t.snd_par, t.snd_child, (int)t.snd_par.size());
  const int Nc = (int)t.snd.size();
  t.sep.assign(Nc, IntSet());
  for (int c = 0; c < Nc; ++c) {
    const int vrep = *t.snd[c].begin();                      // minimum(snode) (trees.jl:497, 518)
    if (vrep + 1 == N) continue;                             // find_higher_order_neighbors returns nothing for the last vertex
    for (int nb : L.cols[vrep])
      if (!t.snd[c].count(nb)) {
        t.sep[c].insert(nb);
        if (strategy == CLIQUE_GRAPH_MERGE) t.snd[c].insert(nb);   // add_separators!: the supernode becomes the full clique
      }
  }
  if (strategy == CLIQUE_GRAPH_MERGE) {                      // trees.jl:93-98: give up the tree structure
    std::fill(t.snd_par.begin(), t.snd_par.end(), -2);
    t.snd_child.assign(Nc, IntSet());
  }
  t.num = Nc;
}

// ---------------------------------------------------------------------------------------------------------------------
// clique_graph.jl
// ---------------------------------------------------------------------------------------------------------------------
static bool is_subset(const IntSet& a, const IntSet& b) { return std::includes(b.begin(), b.end(), a.begin(), a.end()); }
static IntSet intersect(const IntSet& a, const IntSet& b) {
  IntSet r;
  std::set_intersection(a.begin(), a.end(), b.begin(), b.end(), std::inserter(r, r.begin()));
  return r;
}
static int intersect_dim(const IntSet& a, const IntSet& b) {   // clique_merging.jl:601-614
  int d = 0;
  const IntSet& sa = a.size() < b.size() ? a : b;
  const IntSet& sb = a.size() < b.size() ? b : a;
  for (int e : sa) d += (int)sb.count(e);
  return d;
}
static int union_dim(const IntSet& a, const IntSet& b) { return (int)a.size() + (int)b.size() - intersect_dim(a, b); }   // :619-625

// compute_reduced_clique_graph! (clique_graph.jl:17-50; Habib & Stacho 2009).  NB: sorts `sep` in place like the reference.
void compute_reduced_clique_graph(std::vector<IntSet>& sep, const std::vector<IntSet>& snd, std::vector<int>& rows, std::vector<int>& cols) {
  std::stable_sort(sep.begin(), sep.end(), [](const IntSet& a, const IntSet& b) { return a.size() > b.size(); });
  rows.clear(); cols.clear();
  // vertex -> cliques containing it (ascending clique index): the cliques that contain a separator are found among the cliques
  // of its rarest vertex instead of by a scan over all cliques (same set, same ascending order as findall in the reference)
  int maxv = -1;
  for (const IntSet& c : snd) if (!c.empty()) maxv = std::max(maxv, *c.rbegin());
  std::vector<std::vector<int>> occ((size_t)(maxv + 1));
  for (int c = 0; c < (int)snd.size(); ++c) for (int v : snd[c]) occ[(size_t)v].push_back(c);
  for (const IntSet& separator : sep) {
    std::vector<int> clique_ind;
    if (separator.empty()) {
      for (int c = 0; c < (int)snd.size(); ++c) clique_ind.push_back(c);
    } else {
      int vbest = -1;
      for (int v : separator) {
        if (v < 0 || v > maxv) { vbest = -2; break; }                        // a vertex no clique holds: no clique contains the separator
        if (vbest < 0 || occ[(size_t)v].size() < occ[(size_t)vbest].size()) vbest = v;
      }
      if (vbest >= 0) for (int c : occ[(size_t)vbest]) if (is_subset(separator, snd[c])) clique_ind.push_back(c);
    }
    // separator graph H: two cliques are adjacent when their intersection is strictly larger than the separator (:59-87)
    std::map<int, std::vector<int>> H;
    for (int c : clique_ind) H[c];
    for (size_t a = 0; a < clique_ind.size(); ++a)
      for (size_t b = a + 1; b < clique_ind.size(); ++b) {
        const int ca = clique_ind[a], cb = clique_ind[b];
        if (intersect(snd[ca], snd[cb]) != separator) { H[ca].push_back(cb); H[cb].push_back(ca); }   // !inter_equal (:115-135)
      }
    // connected components (:90-112)
    std::map<int, int> comp;
    int ncomp = 0;
    for (int v : clique_ind) {
      if (comp.count(v)) continue;
      std::vector<int> stack{v};
      comp[v] = ncomp;
      while (!stack.empty()) {
        const int u = stack.back(); stack.pop_back();
        for (int w : H[u]) if (!comp.count(w)) { comp[w] = ncomp; stack.push_back(w); }
      }
      ++ncomp;
    }
    for (size_t a = 0; a < clique_ind.size(); ++a)
      for (size_t b = a + 1; b < clique_ind.size(); ++b) {
        const int ca = clique_ind[a], cb = clique_ind[b];
        if (comp[ca] != comp[cb]) { rows.push_back(std::max(ca, cb)); cols.push_back(std::min(ca, cb)); }
      }
  }
}

// ispermissible (clique_graph.jl:153-163)
bool ispermissible(int c1, int c2, const std::map<int, IntSet>& adjacency_table, const std::vector<IntSet>& snd) {
  const IntSet common = intersect(adjacency_table.at(c1), adjacency_table.at(c2));
  for (int nb : common)
    if (intersect(snd[c1], snd[nb]) != intersect(snd[c2], snd[nb])) return false;
  return true;
}

// ---------------------------------------------------------------------------------------------------------------------
// clique_merging.jl
// ---------------------------------------------------------------------------------------------------------------------
static double edge_metric(const IntSet& a, const IntSet& b) {   // ComplexityWeight (:386-398): n1^3 + n2^3 - |a u b|^3
  const double n1 = (double)a.size(), n2 = (double)b.size(), nm = (double)union_dim(a, b);
  return n1 * n1 * n1 + n2 * n2 * n2 - nm * nm * nm;
}

void initialise(SuperNodeTree& t) {
  t.stop = false;
  if (t.strategy == CLIQUE_GRAPH_MERGE) {                     // :219-231
    std::vector<int> rows, cols;
    compute_reduced_clique_graph(t.sep, t.snd, rows, cols);
    t.edges.e.clear();
    for (size_t k = 0; k < rows.size(); ++k) t.edges.e[{cols[k], rows[k]}] += edge_metric(t.snd[rows[k]], t.snd[cols[k]]);   // sparse(): duplicates add
    t.adjacency_table.clear();                               // compute_adjacency_table (clique_graph.jl:138-150)
    for (int i = 0; i < t.num; ++i) t.adjacency_table[i];
    for (auto& kv : t.edges.e) { t.adjacency_table[kv.first.second].insert(kv.first.first); t.adjacency_table[kv.first.first].insert(kv.first.second); }
  } else if (t.strategy == PARENT_CHILD_MERGE) {              // :233-236: start with the clique of second highest order
    t.clique_ind = (int)t.snd.size() - 2;
  }
}

// traverse (:250-275).  Returns false when no candidate exists (empty clique graph).
bool traverse(SuperNodeTree& t, int cand[2]) {
  if (t.strategy == PARENT_CHILD_MERGE) {
    const int c = t.snd_post[t.clique_ind];
    cand[0] = t.snd_par[c]; cand[1] = c;
    return true;
  }
  if (t.edges.e.empty()) return false;
  // max_elem (:405-421): first maximum in CSC order
  auto best = t.edges.e.begin();
  for (auto it = t.edges.e.begin(); it != t.edges.e.end(); ++it) if (it->second > best->second) best = it;
  if (ispermissible(best->first.second, best->first.first, t.adjacency_table, t.snd)) { cand[0] = best->first.second; cand[1] = best->first.first; return true; }
  // weights in decreasing order (the reference uses an unstable QuickSort: ties may be visited in a different order)
  std::vector<std::map<std::pair<int, int>, double>::iterator> its;
  for (auto it = t.edges.e.begin(); it != t.edges.e.end(); ++it) its.push_back(it);
  std::stable_sort(its.begin(), its.end(), [](const auto& a, const auto& b) { return a->second > b->second; });
  for (size_t k = 1; k < its.size(); ++k) {
    const int r = its[k]->first.second, c = its[k]->first.first;
    if (ispermissible(r, c, t.adjacency_table, t.snd)) { cand[0] = r; cand[1] = c; return true; }
  }
  return false;
}

static int fill_in(int dcs, int dcp, int dps, int dpp) { return ((dps + dpp) - dcp) * ((dcs + dcp) - dcp); }   // :628-632

bool evaluate(SuperNodeTree& t, const int cand[2]) {
  if (t.strategy == PARENT_CHILD_MERGE) {                     // :278-286
    if (t.stop) return false;
    const int par = cand[0], c = cand[1];
    const int dps = (int)t.snd[par].size(), dpp = (int)t.sep[par].size(), dcs = (int)t.snd[c].size(), dcp = (int)t.sep[c].size();
    return fill_in(dcs, dcp, dps, dpp) <= t.t_fill || std::max(dcs, dps) <= t.t_size;
  }
  const bool do_merge = t.edges.get(cand[0], cand[1]) >= 0.0;  // :289-296
  if (!do_merge) t.stop = true;
  return do_merge;
}

void merge_two_cliques(SuperNodeTree& t, const int cand[2]) {
  if (t.strategy == PARENT_CHILD_MERGE) {                     // merge_child! (:176-200)
    int p = cand[1], ch = cand[0];
    if (t.snd_child[cand[0]].count(cand[1])) { p = cand[0]; ch = cand[1]; }
    t.snd[p].insert(t.snd[ch].begin(), t.snd[ch].end());
    t.snd[ch].clear(); t.sep[ch].clear();
    for (int g : t.snd_child[ch]) t.snd_par[g] = p;
    t.snd_par[ch] = -2;
    t.snd_child[p].erase(ch);
    t.snd_child[p].insert(t.snd_child[ch].begin(), t.snd_child[ch].end());
    t.snd_child[ch].clear();
    t.num -= 1;
    return;
  }
  const int c1 = cand[0], c2 = cand[1];                       // :203-214
  t.snd[c1].insert(t.snd[c2].begin(), t.snd[c2].end());
  t.snd[c2].clear();
  t.num -= 1;
}

void update_strategy(SuperNodeTree& t, const int cand[2], bool do_merge) {
  if (t.strategy == PARENT_CHILD_MERGE) {                     // :299-307
    if (t.clique_ind == 0) t.stop = true; else t.clique_ind -= 1;
    return;
  }
  if (!do_merge) return;                                      // :310-358
  const int c1 = cand[0], crem = cand[1];
  const IntSet neighbors = t.adjacency_table[c1];
  IntSet new_neighbors;
  for (int x : t.adjacency_table[crem]) if (!neighbors.count(x) && x != c1) new_neighbors.insert(x);
  for (int nb : neighbors)
    if (nb != crem) t.edges.set(std::max(c1, nb), std::min(c1, nb), edge_metric(t.snd[c1], t.snd[nb]));
  for (int nb : new_neighbors) t.edges.set(std::max(c1, nb), std::min(c1, nb), edge_metric(t.snd[c1], t.snd[nb]));
  for (auto& kv : t.edges.e) if (kv.first.first == crem || kv.first.second == crem) kv.second = 0.0;   // edges[crem+1:n, crem] = edges[crem, 1:crem] = 0
  t.edges.dropzeros();
  t.adjacency_table[c1].insert(new_neighbors.begin(), new_neighbors.end());
  for (int nb : new_neighbors) t.adjacency_table[nb].insert(c1);
  t.adjacency_table.erase(crem);
  for (auto& kv : t.adjacency_table) kv.second.erase(crem);
}

static void log_merge(SuperNodeTree& t, bool do_merge, const int cand[2]) {   // :640-645
  t.merge_log.clique_pairs.push_back({cand[0], cand[1]});
  t.merge_log.decisions.push_back(do_merge ? 1 : 0);
  if (do_merge) t.merge_log.num += 1;
}

static void _merge_cliques(SuperNodeTree& t) {                // :115-137
  initialise(t);
  while (!t.stop) {
    int cand[2];
    if (!traverse(t, cand)) break;
    const bool do_merge = evaluate(t, cand);
    if (do_merge) merge_two_cliques(t, cand);
    log_merge(t, do_merge, cand);
    update_strategy(t, cand, do_merge);
    if (t.num == 1) break;
    if (t.stop) break;
  }
}

// clique_tree_from_graph! (:572-598) with clique_intersections! (:458-469), kruskal! (:482-507), determine_parent_cliques!
// (:526-543), split_cliques! (:546-560)
void clique_tree_from_graph(SuperNodeTree& t) {
  for (auto& kv : t.edges.e) kv.second = (double)intersect_dim(t.snd[kv.first.second], t.snd[kv.first.first]);
  {  // Kruskal: maximum weight spanning tree, selected edges are marked with -1
    const int n0 = (int)t.snd.size();
    std::vector<int> uf(n0);
    std::iota(uf.begin(), uf.end(), 0);
    std::function<int(int)> find = [&](int x) { while (uf[x] != x) { uf[x] = uf[uf[x]]; x = uf[x]; } return x; };
    std::vector<std::map<std::pair<int, int>, double>::iterator> its;
    for (auto it = t.edges.e.begin(); it != t.edges.e.end(); ++it) its.push_back(it);
    std::stable_sort(its.begin(), its.end(), [](const auto& a, const auto& b) { return a->second > b->second; });   // sortperm(V, rev = true) is stable
    int found = 0;
    for (auto it : its) {
      const int r = find(it->first.second), c = find(it->first.first);
      if (r != c) {
        uf[r] = c;
        it->second = -1.0;
        if (++found >= t.num - 1) break;
      }
    }
  }
  // root: the first clique that contains the vertex of highest order
  const int v = t.post.back();
  int root = -1;
  for (int k = 0; k < (int)t.snd.size(); ++k) if (t.snd[k].count(v)) { root = k; break; }
  if (root < 0) throw std::runtime_error("clique_tree_from_graph: no clique contains the root vertex");
  t.snd_par[root] = -1;
  // assign_children! (:510-522): neighbours of c = stored entries of row c (columns < c, ascending) then of column c
  std::function<void(int)> assign = [&](int c) {
    std::vector<int> nbs;
    for (auto& kv : t.edges.e) if (kv.first.second == c && kv.first.first < c) nbs.push_back(kv.first.first);
    for (auto& kv : t.edges.e) if (kv.first.first == c) nbs.push_back(kv.first.second);
    for (int nb : nbs)
      if (t.edges.get(std::max(c, nb), std::min(c, nb)) == -1.0 && t.snd_par[c] != nb) {
        t.snd_par[nb] = c;
        t.snd_child[c].insert(nb);
        assign(nb);
      }
  };
  assign(root);
  t.snd_post = post_order(t.snd_par, t.snd_child, t.num);
  t.sep.assign(t.snd.size(), IntSet());
  for (int j = 0; j + 1 < t.num; ++j) {                       // split_cliques!
    const int c = t.snd_post[j], p = t.snd_par[c];
    t.sep[c] = intersect(t.snd[c], t.snd[p]);
    for (int x : t.sep[c]) t.snd[c].erase(x);
  }
  t.clique_tree_recomputed = true;
}

// merge_cliques! (:141-175)
void merge_cliques(SuperNodeTree& t) {
  if (t.strategy == NO_MERGE) return;
  _merge_cliques(t);
  if (t.strategy == PARENT_CHILD_MERGE) {
    t.snd_post = post_order(t.snd_par, t.snd_child, t.num);
    return;
  }
  t.snd_post.clear();
  for (int c = 0; c < (int)t.snd.size(); ++c) if (!t.snd[c].empty()) t.snd_post.push_back(c);
  std::fill(t.snd_par.begin(), t.snd_par.end(), -2);
  if (t.num > 1) clique_tree_from_graph(t);
  t.edges.e.clear(); t.adjacency_table.clear();                // free_clique_graph!
}

// reorder_snd_consecutively! (trees.jl:534-557): renumber the vertices so that every supernode is a consecutive range (post
// order of the cliques); `ordering` maps the new numbers back to the rows/columns of the original matrix.
void reorder_snd_consecutively(SuperNodeTree& t, std::vector<int>& ordering) {
  const int N = (int)t.post.size();
  std::vector<int> p(N, 0);
  int k = 0;
  for (int c : t.snd_post) {
    IntSet renum;
    for (int v : t.snd[c]) { p[k] = v; renum.insert(k); ++k; }        // sorted: IntSet iterates ascending
    t.snd[c] = renum;
  }
  std::vector<int> p_inv(N, -1);
  for (int i = 0; i < k; ++i) p_inv[p[i]] = i;
  for (auto& s : t.sep) { IntSet m; for (int v : s) m.insert(p_inv[v]); s = m; }
  std::vector<int> old = ordering;
  for (int i = 0; i < N; ++i) ordering[i] = old[p[i]];               // permute!(ordering, p)
}

void calculate_block_dimensions(SuperNodeTree& t) {   // clique_merging.jl:178-186 (block sizes in post order)
  t.nBlk.assign(t.num, 0);
  for (int i = 0; i < t.num; ++i) { const int c = t.snd_post[i]; t.nBlk[i] = (int)(t.sep[c].size() + t.snd[c].size()); }
}

// get_clique(sntree, ind) (trees.jl:268-287): clique with post order `ind`
std::vector<int> get_clique(const SuperNodeTree& t, int ind) {
  const int c = t.snd_post[ind];
  IntSet u = t.snd[c];
  u.insert(t.sep[c].begin(), t.sep[c].end());
  return std::vector<int>(u.begin(), u.end());
}

// ---------------------------------------------------------------------------------------------------------------------
// Ordering + symbolic factorisation.  The reference calls QDLDL.qdldl(sparse(pattern), logical = true), whose default
// permutation is SuiteSparse AMD (AMD.jl) -- an external dependency that is not part of the reference tree (PARITY UNPINNED
// for the ordering).  Any elimination ordering yields a valid chordal extension; this is an exact minimum-degree ordering
// (smallest degree in the elimination graph, ties to the smallest index).  Callers that hold the reference's permutation can
// pass it in instead.
// ---------------------------------------------------------------------------------------------------------------------
std::vector<int> minimum_degree_ordering(int N, const std::vector<IntSet>& adj_in) {
  std::vector<IntSet> adj = adj_in;
  std::vector<char> done(N, 0);
  std::vector<int> perm;
  perm.reserve(N);
  std::set<std::pair<int, int>> heap;     // (degree, vertex)
  for (int v = 0; v < N; ++v) heap.insert({(int)adj[v].size(), v});
  while (!heap.empty()) {
    const int v = heap.begin()->second;
    heap.erase(heap.begin());
    done[v] = 1;
    perm.push_back(v);
    std::vector<int> nb(adj[v].begin(), adj[v].end());
    for (int u : nb) { heap.erase({(int)adj[u].size(), u}); adj[u].erase(v); }
    for (size_t a = 0; a < nb.size(); ++a)
      for (size_t b = a + 1; b < nb.size(); ++b) { adj[nb[a]].insert(nb[b]); adj[nb[b]].insert(nb[a]); }
    for (int u : nb) heap.insert({(int)adj[u].size(), u});
    adj[v].clear();
  }
  return perm;
}

// filled column structure of the LDL' factor of the permuted pattern (perm[k] = original index of pivot k)
LPattern symbolic_ldl(int N, const std::vector<IntSet>& adj, const std::vector<int>& perm) {
  std::vector<int> inv(N);
  for (int k = 0; k < N; ++k) inv[perm[k]] = k;
  LPattern L;
  L.n = N;
  L.cols.assign(N, {});
  std::vector<IntSet> st(N);
  for (int k = 0; k < N; ++k)
    for (int u : adj[perm[k]]) { const int j = inv[u]; if (j > k) st[k].insert(j); }
  for (int k = 0; k < N; ++k) {
    L.cols[k].assign(st[k].begin(), st[k].end());
    if (!st[k].empty()) {
      const int p = *st[k].begin();
      for (int j : st[k]) if (j != p) st[p].insert(j);
    }
    st[k].clear();
  }
  return L;
}

}  // namespace chordal
