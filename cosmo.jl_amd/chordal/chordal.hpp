// chordal.hpp -- host-side chordal decomposition front-end of the MI355X COSMO path.
//
// Restates src/chordal_decomposition/*.jl of COSMO.jl v0.8.11 in C++: aggregate sparsity of every PsdConeTriangle,
// chordal extension (fill-reducing ordering + symbolic LDL'), elimination tree, Pothen-Sun supernodes, separators, clique
// merging (NoMerge / ParentChildMerge / CliqueGraphMerge with the reduced clique graph), clique tree from the merged graph
// (Kruskal), the compact clique-tree transformation of (A, b, cones), and the reverse step (block re-assembly, positive
// semidefinite completion of the dual).  It is pure integer / graph code around the hot path: it runs once per solve and
// produces the already-decomposed problem that the device loop iterates on (BASELINE config 5).
//
// Indices are 0-based here (the reference is 1-based); parents use -1 for "root" (reference: 0) and -2 for "removed by a
// merge" (reference: -1).  Julia's `Set` iterates in hash order; wherever the reference iterates a Set this code iterates in
// ascending order, which changes clique numbering but never the set of cliques (documented per function).
#pragma once
#include <array>
#include <cstdint>
#include <map>
#include <set>
#include <string>
#include <utility>
#include <vector>

namespace chordal {

using IntSet = std::set<int>;

enum MergeStrategy { NO_MERGE = 0, PARENT_CHILD_MERGE = 1, CLIQUE_GRAPH_MERGE = 2 };

struct MergeLog {  // src/chordal_decomposition/trees.jl:33-46
  int num = 0;
  std::vector<std::array<int, 2>> clique_pairs;
  std::vector<char> decisions;
};

// strictly lower triangular sparsity pattern of the LDL' factor, column j holds the rows i > j
struct LPattern {
  int n = 0;
  std::vector<std::vector<int>> cols;   // sorted ascending
};

// SparseMatrixCSC{Float64} restricted to what CliqueGraphMerge does with `edges` (lower triangular, row > col): iteration in
// CSC order (column major, rows ascending), insertion on assignment, dropzeros!
struct EdgeMatrix {
  std::map<std::pair<int, int>, double> e;   // key (col, row)
  double get(int row, int col) const { auto it = e.find({col, row}); return it == e.end() ? 0.0 : it->second; }
  bool stored(int row, int col) const { return e.count({col, row}) != 0; }
  void set(int row, int col, double v) {     // setindex!: a zero is only written onto an already stored entry
    auto it = e.find({col, row});
    if (it != e.end()) it->second = v; else if (v != 0.0) e[{col, row}] = v;
  }
  void dropzeros() { for (auto it = e.begin(); it != e.end();) { if (it->second == 0.0) it = e.erase(it); else ++it; } }
};

struct SuperNodeTree {   // src/chordal_decomposition/trees.jl:61-124
  std::vector<IntSet> snd, sep;
  std::vector<int> snd_par;
  std::vector<int> snd_post;
  std::vector<IntSet> snd_child;
  std::vector<int> post;     // post ordering of the vertices of the elimination tree
  std::vector<int> par;      // elimination tree
  std::vector<int> nBlk;
  int num = 0;
  MergeLog merge_log;
  // merge strategy state (clique_merging.jl:62-101)
  int strategy = CLIQUE_GRAPH_MERGE;
  bool stop = false;
  int clique_ind = 0;        // ParentChildMerge
  int t_fill = 8, t_size = 8;
  EdgeMatrix edges;          // CliqueGraphMerge
  std::map<int, IntSet> adjacency_table;
  bool clique_tree_recomputed = false;
  // after merge_cliques!: snd / sep as sorted arrays (graph-based) -- here: vectors in the reference's order
  std::vector<std::vector<int>> snd_v, sep_v;
};

// ---- trees.jl -------------------------------------------------------------------------------------------------------
std::vector<int> etree(const LPattern& L);
std::vector<IntSet> child_from_par(const std::vector<int>& par);
std::vector<int> post_order(const std::vector<int>& par, const std::vector<IntSet>& child, int Nc);
std::vector<int> higher_degrees(const LPattern& L);
void pothen_sun(const std::vector<int>& par, const std::vector<int>& post, const std::vector<int>& degrees, std::vector<int>& sn_par,
                std::vector<int>& sn_ind);
void build_supernode_tree(SuperNodeTree& t, const LPattern& L, int strategy, int t_fill, int t_size);
void connect_graph(LPattern& L);

// ---- clique_graph.jl / clique_merging.jl ------------------------------------------------------------------------------
void compute_reduced_clique_graph(std::vector<IntSet>& sep, const std::vector<IntSet>& snd, std::vector<int>& rows, std::vector<int>& cols);
bool ispermissible(int c1, int c2, const std::map<int, IntSet>& adjacency_table, const std::vector<IntSet>& snd);
void initialise(SuperNodeTree& t);
bool traverse(SuperNodeTree& t, int cand[2]);
bool evaluate(SuperNodeTree& t, const int cand[2]);
void merge_two_cliques(SuperNodeTree& t, const int cand[2]);
void update_strategy(SuperNodeTree& t, const int cand[2], bool do_merge);
void merge_cliques(SuperNodeTree& t);
void clique_tree_from_graph(SuperNodeTree& t);
void reorder_snd_consecutively(SuperNodeTree& t, std::vector<int>& ordering);
void calculate_block_dimensions(SuperNodeTree& t);
std::vector<int> get_clique(const SuperNodeTree& t, int ind);

// ---- ordering + symbolic factorisation (stand-in for QDLDL.qdldl(...; logical = true) with its AMD permutation) ---------
std::vector<int> minimum_degree_ordering(int N, const std::vector<IntSet>& adj);
LPattern symbolic_ldl(int N, const std::vector<IntSet>& adj, const std::vector<int>& perm);

}  // namespace chordal
