/* cosmo_chordal.h -- C ABI of libcosmo_chordal.so: the chordal decomposition front-end of COSMO restated natively (host C++).
 *
 * Replaces chordal_decomposition!(ws) and reverse_decomposition!(ws, settings)
 * (src/chordal_decomposition/chordal_decomposition.jl:10-38, 126-150 of COSMO.jl v0.8.11), i.e. what optimize! runs before
 * setup! and after the loop when settings.decompose = true (src/solver.jl:88-94, 184-190).  The output of
 * cosmo_chordal_decompose is the already-decomposed problem (compact clique-tree transformation, settings
 * .compact_transformation = true: src/chordal_decomposition/transformations.jl:152-200) that is handed to
 * cosmo_hip_set_problem / cosmo_hip_set_cones.
 *
 * Conventions: the INTERNAL problem convention of the reference (A x + s = b, s in K; see src/interface.jl:478-484), CSC with
 * 1-based Int64 indices exactly as Julia's SparseMatrixCSC stores them, cone type codes of cosmo_hip.h.  With the compact
 * transformation only PsdConeTriangle (type 5) cones are decomposed, like the reference's (transformations.jl:276, add_entries!
 * is specialised on PsdConeTriangle{Float64}); the traditional transformation also decomposes PsdCone (type 4); every other
 * cone is passed through.
 * All functions return 0 on success, non-zero on failure (cosmo_chordal_last_error gives the text).
 */
#ifndef COSMO_CHORDAL_H
#define COSMO_CHORDAL_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct cosmo_chordal cosmo_chordal;

enum { COSMO_CHORDAL_NO_MERGE = 0, COSMO_CHORDAL_PARENT_CHILD_MERGE = 1, COSMO_CHORDAL_CLIQUE_GRAPH_MERGE = 2 };

typedef struct cosmo_chordal_options {
  int32_t merge_strategy;  /* settings.merge_strategy: CliqueGraphMerge (default, src/settings.jl:134), ParentChildMerge, NoMerge */
  int32_t t_fill, t_size;  /* ParentChildMerge(t_fill = 8, t_size = 8) (src/chordal_decomposition/clique_merging.jl:91-101) */
  /* Optional elimination orderings, one per PsdConeTriangle cone in cone order, concatenated; ordering[k] = 1-based vertex
   * eliminated k-th (what QDLDL returns as F.perm, src/chordal_decomposition/trees.jl:636-641).  NULL: an exact
   * minimum-degree ordering is computed (stand-in for the external AMD; any ordering gives a valid decomposition). */
  const int64_t* orderings;
  /* settings.compact_transformation (src/settings.jl:135, default true).  0: the "traditional" transformation
   * (find_decomposition_matrix! + augment_system!, src/chordal_decomposition/transformations.jl:4-138): s = H sbar with one
   * ZeroSet(m) block in front; decomposes PsdCone (square) as well as PsdConeTriangle.  With orderings given, one ordering per
   * decomposable cone kind that is active in the chosen mode. */
  int32_t compact_transformation;
} cosmo_chordal_options;

void cosmo_chordal_default_options(cosmo_chordal_options* o);

/* chordal_decomposition!: analyse the sparsity of every PsdConeTriangle, merge cliques, build the decomposed problem. */
int32_t cosmo_chordal_decompose(int64_t n, int64_t m, const int64_t* A_colptr, const int64_t* A_rowval, const double* A_nzval,
                                const double* b, int64_t ncones, const int32_t* type, const int64_t* dim,
                                const cosmo_chordal_options* opt, cosmo_chordal** out);
void cosmo_chordal_free(cosmo_chordal* c);
const char* cosmo_chordal_last_error(void);

/* sizes[0..5] = {n_new, m_new, nnz(A_new), ncones_new, number of decomposed cones, number of overlap columns} */
int32_t cosmo_chordal_sizes(const cosmo_chordal* c, int64_t sizes[6]);
/* the decomposed problem (ws.p.A, ws.p.b, ws.p.C after augment_clique_based!) and cone_map (new cone -> original cone, 1-based,
 * src/types.jl:238).  P and q are extended with zeros by the caller (transformations.jl:193-194).  clique_of[k] = post-order
 * index (1-based) of the clique a decomposed PSD cone k stands for, 0 for pass-through cones. */
int32_t cosmo_chordal_get_problem(const cosmo_chordal* c, int64_t* A_colptr, int64_t* A_rowval, double* A_nzval, double* b, int32_t* type,
                                  int64_t* dim, int64_t* cone_map, int64_t* clique_of);
/* cliques of decomposed original cone `cone` (1-based index among ALL original cones): count, then for clique q (post order)
 * its vertices (1-based rows/columns of the original matrix, ascending) in `vertices` starting at `ptr[q]`. */
int32_t cosmo_chordal_num_cliques(const cosmo_chordal* c, int64_t cone, int64_t* num_cliques, int64_t* total_vertices);
int32_t cosmo_chordal_get_cliques(const cosmo_chordal* c, int64_t cone, int64_t* ptr, int64_t* vertices);
/* merge log of that cone: pairs (1-based clique indices, row-major 2 columns) and decisions */
int32_t cosmo_chordal_merge_log(const cosmo_chordal* c, int64_t cone, int64_t* num_decisions, int64_t* num_merges, int64_t* pairs, int32_t* decisions,
                                int64_t capacity);

/* reverse_decomposition! for the compact transformation: add_sub_blocks! (chordal_decomposition.jl:170-215) and, when
 * complete_dual != 0, psd_completion! of y = -mu (:220-311).  Inputs: decomposed s, mu (length m_new); outputs: s, mu of the
 * original problem (length m). */
int32_t cosmo_chordal_reverse(const cosmo_chordal* c, const double* s_dec, const double* mu_dec, double* s_out, double* mu_out, int32_t complete_dual);

/* ---- hooks used by the golden tests of the merging machinery (test/UnitTests/DecompositionTests) -------------------- */
/* clique tree given explicitly (1-based): snd/sep as concatenated sets with pointers, parents (0 = root), post order.  Runs
 * merge_cliques! with the strategy and returns the log and the final parents. */
int32_t cosmo_chordal_test_merge_tree(int64_t ncliques, const int64_t* snd_ptr, const int64_t* snd, const int64_t* sep_ptr, const int64_t* sep,
                                      const int64_t* par, const int64_t* snd_post, int64_t nvertices, int32_t strategy, int64_t* num_decisions,
                                      int64_t* num_merges, int64_t* pairs, int32_t* decisions, int64_t* par_out, int64_t capacity);
/* compute_reduced_clique_graph!(sep, snd) (clique_graph.jl:17-50): edges as (row > col), 1-based; plus the ComplexityWeight
 * of every edge and whether it is permissible (clique_graph.jl:153-163) */
int32_t cosmo_chordal_test_reduced_clique_graph(int64_t ncliques, const int64_t* snd_ptr, const int64_t* snd, int64_t nsep, const int64_t* sep_ptr,
                                                const int64_t* sep, int64_t* nedges, int64_t* rows, int64_t* cols, double* weights, int32_t* permissible,
                                                int64_t capacity);
#ifdef __cplusplus
}
#endif
#endif
