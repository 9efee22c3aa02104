/* igcn.h — C ABI of libigcn.so: hand-written gfx950 (CDNA4) kernels for IG-GCN's hot path.
 *
 * The reference (Houliang-Zhou/IG-GCN) is pure Python with no FFI: the path is entered through
 * kernel/sgcn_img_snp.py (SGCN_GCN_IMGSNP) and kernel/go_model.py (Gene_ontology_network), whose sparse
 * arithmetic lives in PyG 2.0.2 / torch-scatter 2.0.9 / torch.sparse.  Each entry point below names the
 * reference operator call site(s) it replaces (paths relative to the reference repo).
 *
 * Conventions
 *  - every pointer is a CALLER-OWNED DEVICE pointer (contiguous, from tensor.data_ptr()); the library
 *    never allocates, frees, copies to host or synchronises; all work is enqueued on `stream`
 *    (a hipStream_t passed as void*), so every call may be captured into a hipGraph.
 *  - float = fp32.  edge_index stays int64 at the boundary (PyG surface, sgcn_data.py:268-269); the plan
 *    holds int32 copies/permutations for the kernels.
 *  - return 0 on success, a negative IGCN_ERR_* otherwise; igcn_last_error() gives a thread-local text.
 *  - "CSR" = (ptr[n_rows+1], idx[nnz]) int32.  GO activations are CHANNEL-MAJOR: [B, f, N] (node index
 *    fastest) so that every access with consecutive nodes on consecutive lanes is coalesced.
 */
#ifndef IGCN_H
#define IGCN_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define IGCN_OK 0
#define IGCN_ERR_BADARG (-1)
#define IGCN_ERR_LAUNCH (-2)
#define IGCN_ERR_UNSUPPORTED (-3)

/* ABI revision: bumped whenever a prototype below changes (argument added, removed or re-ordered).  igcn_version()
 * returns the revision the library was built from; a binding written against a different one must refuse to call
 * (igcn_amd/_lib.py does). */
#define IGCN_ABI_VERSION 423
int igcn_version(void);
const char* igcn_last_error(void);
/* A/B switches, set ONCE by the binding when it loads the library (from the IGCN_* environment variables): bit 0 no tiled
 * list walks, 1 no LDS-staged dense aggregation, 2 untiled CSR map kernels, 3 global-memory GO attention backward, 4 print
 * deferred reductions, 5 fp32 attention core under bf16 transforms; gemm_bn_cap > 0 caps the GEMM tile width, attn_chunk_rows > 0 the rows per LDS chunk of the streamed
 * attention kernels (sweeps).  No launch path reads the environment. */
int igcn_configure(unsigned options, int gemm_bn_cap, int attn_chunk_rows);

/* ------------------------------------------------------------------------------------------------
 * Graph plan: replaces the index work PyG's gcn_norm/propagate redo in every GCNConv call
 * (mask src!=dst, cat, arange, scatter index math; call sites kernel/sgcn_img_snp.py:218,221).
 * Built once per batch and shared by both forward passes, loss_probability and the backward.
 *   src32/dst32 [E]   int32 copies of edge_index rows
 *   tgt_ptr [N+1], tgt_perm [E] : edges grouped by TARGET node, original edge order kept inside a
 *                                 group (stable) => per-target sums run in the reference's scatter order
 *   src_ptr [N+1], src_perm [E] : same, grouped by SOURCE node (transposed structure for the backward)
 *   loop_edge [N]               : id of the LAST stored self-loop of node i, or -1
 * workspace: igcn_graph_plan_workspace_bytes(n_nodes, n_edges) bytes.
 */
size_t igcn_graph_plan_workspace_bytes(int64_t n_nodes, int64_t n_edges);
int igcn_graph_plan_build(int64_t n_nodes, int64_t n_edges, const int64_t* edge_index /*[2,E]*/,
                          int32_t* src32, int32_t* dst32, int32_t* tgt_ptr, int32_t* tgt_perm,
                          int32_t* src_ptr, int32_t* src_perm, int32_t* loop_edge,
                          void* workspace, size_t workspace_bytes, void* stream);

/* Same plan for a block-diagonal PyG batch of SMALL graphs, one workgroup per graph, everything in LDS (no
 * device-wide sort): graph g owns nodes [node_ptr[g],node_ptr[g+1]) and edges [edge_ptr[g],edge_ptr[g+1]) (int64
 * device arrays of n_graphs+1 entries, as Batch.from_data_list records them; batch.py:24-123).  Limits: 1024 nodes
 * and 4096 edges per graph (IGCN_ERR_UNSUPPORTED otherwise: use igcn_graph_plan_build_tiled / igcn_graph_plan_build).
 * `status` (device int32, zeroed by the caller) is set non-zero if a graph exceeds the limits or an edge leaves its
 * graph. */
int igcn_graph_plan_build_segmented(int64_t n_nodes, int64_t n_edges, int n_graphs, const int64_t* edge_index,
                                    const int64_t* node_ptr, const int64_t* edge_ptr,
                                    int64_t max_nodes_per_graph, int64_t max_edges_per_graph,
                                    int32_t* src32, int32_t* dst32, int32_t* tgt_ptr, int32_t* tgt_perm,
                                    int32_t* src_ptr, int32_t* src_perm, int32_t* loop_edge, int32_t* status,
                                    void* stream);
/* The same build, also filling the plan of `copies` disjoint copies of the batch (the seven arrays
 * igcn_graph_plan_replicate derives: copies*E, copies*E, copies*N+1, copies*E, copies*N+1, copies*E, copies*N). */
int igcn_graph_plan_build_segmented_rep(int64_t n_nodes, int64_t n_edges, int n_graphs, const int64_t* edge_index,
                                        const int64_t* node_ptr, const int64_t* edge_ptr, int64_t max_nodes_per_graph,
                                        int64_t max_edges_per_graph, int32_t* src32, int32_t* dst32, int32_t* tgt_ptr,
                                        int32_t* tgt_perm, int32_t* src_ptr, int32_t* src_perm, int32_t* loop_edge,
                                        int32_t* status, int copies, int32_t* o_src32, int32_t* o_dst32,
                                        int32_t* o_tgt_ptr, int32_t* o_tgt_perm, int32_t* o_src_ptr, int32_t* o_src_perm,
                                        int32_t* o_loop_edge, void* stream);

/* Same plan for a PyG batch whose graphs have at most 1024 nodes and ANY number of edges (the dense 512-ROI graphs of
 * the stress configuration: 262 144 edges each): a hand-written one-pass stable counting sort per graph over tiles of
 * 4096 edges (histogram / scan / ranked scatter, three launches for both groupings), the node id inside its graph
 * being the digit.  Same outputs and `status` convention as the segmented build; workspace:
 * igcn_graph_plan_tiled_workspace_bytes(n_graphs, max_nodes_per_graph, max_edges_per_graph) bytes. */
size_t igcn_graph_plan_tiled_workspace_bytes(int n_graphs, int64_t max_nodes_per_graph, int64_t max_edges_per_graph);
int igcn_graph_plan_build_tiled(int64_t n_nodes, int64_t n_edges, int n_graphs, const int64_t* edge_index,
                                const int64_t* node_ptr, const int64_t* edge_ptr,
                                int64_t max_nodes_per_graph, int64_t max_edges_per_graph,
                                int32_t* src32, int32_t* dst32, int32_t* tgt_ptr, int32_t* tgt_perm,
                                int32_t* src_ptr, int32_t* src_perm, int32_t* loop_edge, int32_t* status,
                                void* workspace, size_t workspace_bytes, void* stream);

/* Plan of `copies` disjoint copies of the batch (node g*N+i, edge g*E+k) derived from an existing plan without
 * sorting again; output arrays are sized for copies*N nodes / copies*E edges. */
int igcn_graph_plan_replicate(int64_t n_nodes, int64_t n_edges, int copies,
                              const int32_t* src32, const int32_t* dst32, const int32_t* tgt_ptr,
                              const int32_t* tgt_perm, const int32_t* src_ptr, const int32_t* src_perm,
                              const int32_t* loop_edge,
                              int32_t* o_src32, int32_t* o_dst32, int32_t* o_tgt_ptr, int32_t* o_tgt_perm,
                              int32_t* o_src_ptr, int32_t* o_src_perm, int32_t* o_loop_edge, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Learned regional / connective importance masks — cal_probability, kernel/sgcn_img_snp.py:133-151.
 *   xm[i,:]  = x[i,:] * prob[i % rois,:]
 *   e[k]     = sigmoid( [xm[src_k] || xm[dst_k]] . prob_bias )          ewm[k] = ew[k] * e[k]
 * x_plain [N,h0] / ew_plain [E] (may be NULL): plain copies of x / ew written alongside — the train step stacks the
 * plain and the masked pass (train() :521,523) into one batch, and xm / x_plain (ewm / ew_plain) are then the two
 * halves of the stacked tensors.
 */
int igcn_edge_mask_fwd(int64_t n_nodes, int64_t n_edges, int rois, int h0,
                       const float* x, const float* prob, const float* prob_bias, const float* ew,
                       const int32_t* src32, const int32_t* dst32,
                       float* xm, float* e, float* ewm, float* x_plain, float* ew_plain, void* stream);
/* d_xm [N,h0] may be NULL (=0); d_ewm, d_e [E] may be NULL; d_x_plain [N,h0] (may be NULL) is the gradient of the
 * plain copy, added into dx.  Outputs: dx [N,h0], dprob [rois,h0], dprob_bias [2*h0].
 * scratch: float[ N*h0 + 16*ceil(N/4) + 16 ]. */
int igcn_edge_mask_bwd(int64_t n_nodes, int64_t n_edges, int rois, int h0,
                       const float* x, const float* prob, const float* prob_bias, const float* ew,
                       const float* e, const float* d_xm, const float* d_ewm, const float* d_e,
                       const float* d_x_plain, const int32_t* tgt_ptr, const int32_t* tgt_perm,
                       const int32_t* src_ptr, const int32_t* src_perm,
                       float* dx, float* dprob, float* dprob_bias, float* scratch, void* stream);

/* The same two entry points with loss_probability (kernel/sgcn_img_snp.py:153-181) riding along — in a train step the
 * regulariser reads exactly what the mask launch has in registers (e per edge, prob, the SNP mask logits):
 *   forward: reg_partial [igcn_edge_mask_reg_blocks(...)] workgroup partials whose SUM is
 *            mean r_x(sigmoid(prob)) + mean r_e(e) + mean r_x(sigmoid(snps_logits))     (snps_logits NULL: without it)
 *   backward: d_reg [1] (device) = d loss / d (that sum); its edge part joins d_e inside the node pass, its prob part
 *            dprob, and dsnps [n_snps] receives the SNP logits' part.
 * Same arithmetic as igcn_mask_reg_{fwd,bwd}; two launches less per step. */
int igcn_edge_mask_reg_blocks(int64_t n_nodes, int64_t n_edges, int h0, int n_snps, int snps_rows);
/* snps_feat [snps_rows, n_snps] (or NULL): the launch also writes the SNP mask of the stacked sweep (cal_probability
 * :147-151), snps_full [2 snps_rows, n_snps] = (snps_feat | snps_feat * sigmoid(snps_logits)); the backward then takes
 * d_snps_full and adds sum_b d_masked[b,k] snps_feat[b,k] sigmoid'(logit_k) to dsnps — four launches of a train step
 * (mask, SNP mask, regulariser, and their backward pair) become the mask's two. */
int igcn_edge_mask_fwd_reg(int64_t n_nodes, int64_t n_edges, int rois, int h0, const float* x, const float* prob,
                           const float* prob_bias, const float* ew, const int32_t* src32, const int32_t* dst32, float* xm,
                           float* e, float* ewm, float* x_plain, float* ew_plain, const float* snps_logits, int n_snps,
                           float l1_x, float ent_x, float l1_e, float ent_e, float eps, float* reg_partial,
                           const float* snps_feat, int snps_rows, float* snps_full, void* stream);
int igcn_edge_mask_bwd_reg(int64_t n_nodes, int64_t n_edges, int rois, int h0, const float* x, const float* prob,
                           const float* prob_bias, const float* ew, const float* e, const float* d_xm, const float* d_ewm,
                           const float* d_e, const float* d_x_plain, const int32_t* tgt_ptr, const int32_t* tgt_perm,
                           const int32_t* src_ptr, const int32_t* src_perm, const float* d_reg, const float* snps_logits,
                           int n_snps, float l1_x, float ent_x, float l1_e, float ent_e, float eps, float* dx, float* dprob,
                           float* dprob_bias, float* dsnps, float* scratch, const float* snps_feat, int snps_rows,
                           const float* d_snps_full, void* stream);

/* ------------------------------------------------------------------------------------------------
 * GCN normalisation — PyG gcn_norm inside GCNConv (kernel/sgcn_img_snp.py:218,221):
 * stored self-loops are replaced by one loop per node (weight wl = last stored loop weight, else 1),
 *   deg[i] = sum_{k: dst=i, src!=dst} ew[k] + wl[i] ;  dis = deg^-1/2 (inf -> 0)
 *   what[k] = dis[src]*ew[k]*dis[dst]  (0 at stored loops) ;  what_loop[i] = dis[i]*wl[i]*dis[i]
 * Computed ONCE per forward pass (the reference recomputes it in every layer); the L layers and the
 * backward all read the same coefficient arrays.
 */
int igcn_gcn_norm_fwd(int64_t n_nodes, int64_t n_edges, const float* ew,
                      const int32_t* src32, const int32_t* dst32, const int32_t* tgt_ptr, const int32_t* tgt_perm,
                      const int32_t* src_perm, const int32_t* loop_edge, float* dis, float* wl,
                      float* what, float* what_loop, void* tstream /*[E] x 8 B*/, void* sstream /*[E] x 8 B*/,
                      void* stream);
/* tstream[p] = {int32 source node, float coefficient} of the p-th edge of the by-TARGET grouping, sstream[p] =
 * {int32 target node, float coefficient} of the p-th edge of the by-SOURCE grouping: the aggregation kernels stream
 * these 8-byte records coalesced instead of chasing permutation -> edge -> endpoint. */
/* dwhat [E] (entries of stored loops ignored) and dwhat_loop [N] are d(loss)/d(coefficient) summed over all
 * layers.  Output dew [E].  scratch: float[N]. */
int igcn_gcn_norm_bwd(int64_t n_nodes, int64_t n_edges, const float* ew, const float* dis, const float* wl,
                      const float* dwhat, const float* dwhat_loop,
                      const int32_t* src32, const int32_t* dst32,
                      const int32_t* tgt_ptr, const int32_t* tgt_perm,
                      const int32_t* src_ptr, const int32_t* src_perm, const int32_t* loop_edge,
                      float* dew, float* scratch, void* stream);

/* ------------------------------------------------------------------------------------------------
 * GCN scatter-aggregate (the north-star kernel) — GCNConv.propagate + bias (+ the F.relu of
 * kernel/sgcn_img_snp.py:218,221):
 *   out[t, :F] = act( sum_{p in group(t)} tstream[p].w * h[tstream[p].idx,:] + what_loop[t]*h[t,:] + bias )
 * summed in the reference's scatter order (non-loop edges in stored order, then the loop).
 * h [N,F] row stride ld_h; out row stride ld_out (so a layer can write its column slice of the JK-concat
 * buffer of :223 directly).  relu != 0 applies max(.,0).  Three launch shapes, picked from the arguments:
 * thread = (target, feature quad) for low in-degree (brain graphs, k = 3); for an average in-degree >= 16 one wave per
 * target in which every lane moves 16 bytes per memory instruction (F/4 lanes per feature row, 64/(F/4) edge slots);
 * and, when `nodes_per_graph` (uniform graph size of a block-diagonal batch, 0 = unknown) is given, the average
 * in-degree is >= 64 and a graph's rows fit 64 KB of LDS (the dense 512-ROI graphs of the stress configuration), a
 * workgroup per 64 targets of one graph that stages the graph's rows in LDS and streams the 8-byte records from HBM
 * (two per 16-byte load) — per-target sums then run lane-parallel, not in the sequential scatter order.
 */
int igcn_gcn_propagate_fwd(int64_t n_nodes, int64_t n_edges, int F, int nodes_per_graph /*0 = unknown*/,
                           const float* h, int64_t ld_h, const void* tstream, const float* what_loop, const float* bias,
                           const int32_t* tgt_ptr, float* out, int64_t ld_out, int relu, void* stream);
/* Backward of the above.  g = dout * (out>0 if relu).  Outputs:
 *   dh [N,F] (row stride ld_dh)  = A_hat^T g          dbias [F] (may be NULL)
 *   dwhat [E] = g[dst_k].h[src_k] (0 at stored loops), dwhat_loop [N] = g[i].h[i]   (only when need_dw != 0)
 * scratch: igcn_gcn_propagate_bwd_scratch_floats(n_nodes, F) floats (block partials of dbias + the ReLU-masked
 * copy of dout that the per-edge kernels gather). */
size_t igcn_gcn_propagate_bwd_scratch_floats(int64_t n_nodes, int F);
int igcn_gcn_propagate_bwd(int64_t n_nodes, int64_t n_edges, int F, int nodes_per_graph /*0 = unknown*/,
                           const float* dout, int64_t ld_dout, const float* out, int64_t ld_out, int relu,
                           const float* h, int64_t ld_h, const void* sstream, const float* what_loop,
                           const int32_t* src32, const int32_t* dst32, const int32_t* src_ptr,
                           float* dh, int64_t ld_dh, float* dbias,
                           int need_dw, float* dwhat, float* dwhat_loop,
                           float* scratch, void* stream);

/* ------------------------------------------------------------------------------------------------
 * The whole SGCN stack of kernel/sgcn_img_snp.py:218-224 for batches of SMALL UNIFORM graphs, LDS-resident: PyG
 * gcn_norm + L x (X W_l^T, scatter-aggregate, + b_l, ReLU) + the jumping-knowledge concatenation as ONE kernel per
 * direction, one workgroup per graph (north_star: "LDS staging of per-block node features").  HBM carries only the
 * compulsory traffic of SURVEY §8d's "fused SGCN forward lower bound": x_in [n_graphs*R, H0], the graph's edges and
 * ew_in in, xcat [n_graphs*R, L*F] out.  Needs a block-diagonal batch (graph g = nodes [gR, (g+1)R), its edges
 * contiguous in stored order — what the per-graph plan builders verify), at most `max_edges` edges per graph,
 * F in {4,8,16,32}, L <= 4, H0 <= 8 and igcn_sgcn_stack_lds_bytes(...) <= 150 KB (IGCN_ERR_UNSUPPORTED otherwise: use
 * igcn_gcn_norm_* / igcn_gemm_f32 / igcn_gcn_propagate_*).  W / b: HOST arrays of L device pointers, W_l [F, Fin_l]
 * row-major (Fin_0 = H0, then F), b_l [F].
 * Backward recomputes the forward in LDS; outputs dx_in [N, H0], dew_in [E] and dparams
 * [igcn_sgcn_stack_param_floats] = dW_0 | db_0 | dW_1 | db_1 | ... (scratch: n_graphs * that many floats; the sum over
 * graphs is a final reduction in the sense of igcn_reduce_defer).  dxcat2 (or NULL): the gradient of a second consumer
 * of xcat, added to dxcat while the rows are staged (the attention query and the head inputs both read xcat). */
size_t igcn_sgcn_stack_lds_bytes(int R, int max_edges, int H0, int F, int L, int backward);
int igcn_sgcn_stack_param_floats(int H0, int F, int L);
int igcn_sgcn_stack_fwd(int64_t n_graphs, int R, int max_edges, int H0, int F, int L, const float* x_in,
                        const float* ew_in, const int32_t* src32, const int32_t* dst32, const int32_t* tgt_ptr,
                        const int32_t* tgt_perm, const int32_t* loop_edge, const float* const* W,
                        const float* const* b, float* xcat, int32_t* status, void* stream);
int igcn_sgcn_stack_bwd(int64_t n_graphs, int R, int max_edges, int H0, int F, int L, const float* x_in,
                        const float* ew_in, const int32_t* src32, const int32_t* dst32, const int32_t* tgt_ptr,
                        const int32_t* tgt_perm, const int32_t* src_ptr, const int32_t* src_perm,
                        const int32_t* loop_edge, const float* const* W, const float* const* b, const float* dxcat,
                        const float* dxcat2, float* dx_in, float* dew_in, float* dparams, float* scratch,
                        int32_t* status, void* stream);
/* THE FRONT OF THE IMAGE BRANCH OF A TRAIN STEP AS ONE LAUNCH (kernel/sgcn_img_snp.py:133-151 cal_probability, :153-181
 * loss_probability, :218-224 the GCNConv stack; train() :521-523 runs them for the plain and the masked pass): for a batch
 * of n_graphs uniform graphs of R nodes (at most max_edges edges each, edge_index int64 [2, n_edges] with per-graph
 * offsets node_ptr / edge_ptr [n_graphs + 1]) workgroup (copy, graph) of the stacked (plain | masked) batch builds the
 * graph's plan in LDS — what igcn_graph_plan_build_segmented_rep(copies = 2) writes: `plan` = the seven arrays of the
 * batch, `plan2` = those of its 2-copy replica — forms the pass's inputs (x_in [2N, H0], ew_in [2E], the edge mask e [E]
 * = igcn_edge_mask_fwd_reg's outputs, reg_partial [n_graphs] whose SUM is loss_probability, snps_full [2 n_graphs, n_snps]
 * = (snps_feat | snps_feat * sigmoid(snps_logits)); snps_feat NULL: no SNP mask) and runs igcn_sgcn_stack_fwd's layers:
 * xcat [2N, L F].  Same arithmetic, in the same order, as the three launches it replaces (the regulariser's partial sums
 * are grouped per graph instead of per 256 edges).  A dropout rider waiting on the stream (igcn_rider_dropout) is carried
 * by this launch.  status: bit 0 = the batch is not block diagonal in the declared segments / not uniform, bit 1 = a graph
 * has more than max_edges edges (its outputs are not written). */
size_t igcn_sgcn_front_lds_bytes(int R, int max_edges, int H0, int F, int L);
int igcn_sgcn_front_fwd(int64_t n_nodes, int64_t n_edges, int n_graphs, int R, int max_edges, int H0, int F, int L,
                        const int64_t* edge_index, const int64_t* node_ptr, const int64_t* edge_ptr,
                        int32_t* const* plan /*HOST [7]*/, int32_t* const* plan2 /*HOST [7]*/, int32_t* status,
                        const float* x, const float* prob, const float* prob_bias, const float* ew, float* x_in,
                        float* ew_in, float* e, const float* snps_logits, int n_snps, float l1_x, float ent_x, float l1_e,
                        float ent_e, float eps, float* reg_partial, const float* snps_feat, float* snps_full,
                        const float* const* W /*HOST [L]*/, const float* const* b /*HOST [L]*/, float* xcat, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Backward of a bias-free projection y = x W^T (x [M, K], W [N, K]) in ONE pass over the gradient G = dL/dy [M, N]:
 * dX [M, K] = G W and dW [N, K] = G^T X — the packed input projection of nn.MultiheadAttention
 * (kernel/sgcn_img_snp.py:240-241), whose key | value gradient is a 52 MB tensor that two separate GEMMs read twice.
 * K == 32, N in {32, 64} (igcn_proj_bwd_supported); scratch: igcn_proj_bwd_scratch_floats(M, N) floats; dW is a final
 * reduction in the sense of igcn_reduce_defer. */
int igcn_proj_bwd_supported(int64_t M, int N, int K);
int igcn_proj_bwd_blocks(int64_t M);
size_t igcn_proj_bwd_scratch_floats(int64_t M, int N);   /* per projection, for every igcn_proj_bwd* entry point */
int igcn_proj_bwd(int64_t M, int N, int K, const float* G, const float* X, const float* W, float* dX, float* dW,
                  float* scratch, void* stream);
/* two projections with the same K in ONE launch (the query block and the key | value block of the packed in-projection) */
int igcn_proj_bwd_pair(int64_t M1, int N1, const float* G1, const float* X1, const float* W1, float* dX1, float* dW1,
                       float* scratch1, int64_t M2, int N2, const float* G2, const float* X2, const float* W2,
                       float* dX2, float* dW2, float* scratch2, int K, void* stream);
/* ... with the bias gradients db_i [N_i] = column sums of G_i from the same pass (NULL: not wanted); the first db_zero_i
 * entries are written as exact zeros (the attention's key bias).  scratch_i: igcn_proj_bwd_scratch_floats(M_i, N_i)
 * floats; db_i are final reductions like dW_i. */
int igcn_proj_bwd_pair_bias(int64_t M1, int N1, const float* G1, const float* X1, const float* W1, float* dX1, float* dW1,
                            float* scratch1, float* db1, int db_zero1, int64_t M2, int N2, const float* G2,
                            const float* X2, const float* W2, float* dX2, float* dW2, float* scratch2, float* db2,
                            int db_zero2, int K, void* stream);
/* Forward of the same projections, y = x W^T + b (bias [N] or NULL), as ONE streaming launch for up to two of them
 * (M2 = 0: the first alone): W and b resident in LDS, x read once, y written in contiguous runs — the in-projection's
 * reduction depth is 32, so the general tiled GEMM's K loop and per-column-tile re-reads of x only cost.  Same shape
 * limits as igcn_proj_bwd_supported; exact fp32. */
int igcn_proj_fwd_blocks(int64_t M);
int igcn_proj_fwd_pair(int64_t M1, int N1, const float* X1, const float* W1, const float* b1, float* Y1, int64_t M2, int N2,
                       const float* X2, const float* W2, const float* b2, float* Y2, int K, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Backward of the two heads' first layers y = relu(x W^T + b) (kernel/sgcn_img_snp.py:299 lin1, :302 lin1_regr; x [R, C]
 * the head inputs, W [64, C]) in one pass over the wide operands: g = dy * (y > 0), db = column sums of g, dx = g W,
 * dW = g^T x — x and W read once, dx written once, no masked copy of dy, no separate bias-gradient launch.
 * Hidden width H == 64, C even (igcn_head_bwd_supported); y_i NULL = no ReLU; C2 = 0 = the first layer alone.  With
 * R > 128 the rows are split: dW_i / db_i are then final reductions in the sense of igcn_reduce_defer over partials in
 * scratch_i (igcn_head_bwd_scratch_floats(R, C_i) floats; must stay alive until the flush).  Exact fp32. */
int igcn_head_bwd_supported(int R, int H, int C);
size_t igcn_head_bwd_scratch_floats(int R, int C);
int igcn_head_bwd_pair(int R, int H, int C1, const float* dy1, const float* y1, const float* W1, const float* X1,
                       float* dX1, float* dW1, float* db1, float* scratch1, int C2, const float* dy2, const float* y2,
                       const float* W2, const float* X2, float* dX2, float* dW2, float* db2, float* scratch2,
                       void* stream);

/* ------------------------------------------------------------------------------------------------
 * SGCN over DENSE brain graphs (BASELINE configs[4]): batches whose graphs are COMPLETE — all R x R (source, target)
 * pairs stored in row-major order, i.e. a dense adjacency turned into COO — so that edge_attr IS the matrix
 * ew[g][src][dst] and edge_index carries no information.  Replaces, for such batches, cal_probability
 * (kernel/sgcn_img_snp.py:133-151), PyG gcn_norm + GCNConv x L + ReLU + concatenation (:218-224) and the edge part of
 * loss_probability (:153-181) of BOTH passes of a train step: every edge pass reads the 4-byte weight once and
 * recomputes mask / coefficient on the fly; the aggregations run on the matrix cores (csrc/sgcn_dense.hip).
 *   igcn_dense_blocks_check : verifies edge k of graph g == (g R + k / R, g R + k % R) — a 16-byte-per-edge read stream.
 *                             status: device int32[2]; status[1] = the LATEST check's verdict (cleared in front of every
 *                             check, 1 on a mismatch), status[0] |= 4 on a mismatch (sticky; a host-side check reads it).
 *   igcn_dense_sgcn_fwd(status, check_edge_index): status = those words, or NULL.  A set verdict turns the first layer's
 *                             features — hence every output and the loss — into NaN (a batch that is not row-major
 *                             complete must not train silently).  check_edge_index != NULL: the check of THIS batch
 *                             rides in the forward's first edge pass (extra workgroups of that launch stream the
 *                             index pairs) instead of a launch of its own in front; NULL: the verdict of an earlier
 *                             igcn_dense_blocks_check on the stream is consumed.
 *   igcn_dense_sgcn_supported: 64 <= R <= 1024, R % 64 == 0, F == 16, L <= 4, H0 <= 8.
 *   copies = 1: one pass (first_masked = isExplain); copies = 2: rows [0, G R) of xcat = plain pass, [G R, 2 G R) = masked.
 *   W / b: HOST arrays of L device pointers (W_l [F, Fin_l] row-major, Fin_0 = H0).  ws: igcn_dense_sgcn_ws_floats
 *   floats, written by fwd and read by bwd.  reg_partials [igcn_dense_sgcn_reg_blocks]: un-reduced partials of
 *   loss_probability (their SUM is the loss; NULL or copies == 1 plain: not written); snps_prob may be NULL.
 *   bwd: dxcat2 = the gradient of a second consumer of xcat (attention query | head inputs), added to dxcat where it is
 *   read, or NULL; d_reg = d loss / d (every partial), one device float (NULL: 0); outputs dx [G R, H0] (both passes' sum),
 *   dprob [R, H0], dprob_bias [2 H0], dsnps_prob [n_snps] (regulariser part only; may be NULL) and dparams
 *   [igcn_sgcn_stack_param_floats] = dW_0 | db_0 | dW_1 | ... (final reductions in the sense of igcn_reduce_defer);
 *   bws: igcn_dense_sgcn_bwd_ws_floats floats of scratch. */
int igcn_dense_blocks_check(int64_t n_graphs, int R, const int64_t* edge_index, int32_t* status, void* stream);
int igcn_dense_sgcn_supported(int R, int H0, int F, int L);
size_t igcn_dense_sgcn_ws_floats(int64_t n_graphs, int R, int L, int copies);
size_t igcn_dense_sgcn_bwd_ws_floats(int64_t n_graphs, int R, int H0, int L, int copies);
int igcn_dense_sgcn_reg_blocks(int64_t n_graphs, int R);
int igcn_dense_sgcn_fwd(int64_t n_graphs, int R, int H0, int F, int L, int copies, int first_masked, const float* x,
                        const float* prob, const float* prob_bias, const float* ew, const float* const* W,
                        const float* const* b, const float* snps_prob, int n_snps, float l1_x, float ent_x, float l1_e,
                        float ent_e, float eps, float* xcat, float* reg_partials, float* ws, int32_t* status,
                        const int64_t* check_edge_index, void* stream);
int igcn_dense_sgcn_bwd(int64_t n_graphs, int R, int H0, int F, int L, int copies, int first_masked, const float* x,
                        const float* prob, const float* prob_bias, const float* ew, const float* const* W,
                        const float* snps_prob, int n_snps, float l1_x, float ent_x, float l1_e, float ent_e, float eps,
                        const float* xcat, const float* dxcat, const float* dxcat2, const float* d_reg, const float* ws,
                        float* bws, float* dx, float* dprob, float* dprob_bias, float* dsnps_prob, float* dparams,
                        void* stream);

/* ------------------------------------------------------------------------------------------------
 * Dense feature transform on the matrix cores (f32-input MFMA 16x16x4, exact fp32):
 *   out[M,N] = act( A[M,K] . W[N,K]^T + bias )      — GCNConv.lin (kernel/sgcn_img_snp.py:34-49),
 *   lin1 / lin1_regr / lin2 / lin2_regr (:62-84,289-301).  General strided form so the same kernel
 *   serves forward (NT), input-gradient (NN) and weight-gradient (TN, split over the long M axis):
 *   C[m,n] = sum_k A[m*sam + k*sak] * B[n*sbn + k*sbk]  (+ bias[n]) ; C row stride ldc.
 * split_k > 1 writes split_k partial slabs into `scratch` (float[split_k*M*N]) and reduces them in order
 * (deterministic).  act: 0 none, 1 relu.
 */
/* Narrow output layer y[R,C] = x[R,K] W[C,K]^T + b[C] with C <= 4 (lin2 / lin2_regr, kernel/sgcn_img_snp.py:289-301)
 * as one VALU kernel per direction.  K/4 must be a power of two <= 64; x, W, dx 16-byte aligned.
 * Backward: dx [R,K] (may be NULL) and dwb [C*K + C] = dW (row-major [C,K]) followed by db.
 * scratch: igcn_small_linear_bwd_scratch_floats(R, K, C). */
size_t igcn_small_linear_bwd_scratch_floats(int64_t R, int K, int C);
/* `keep` [R,K] (may be NULL): dropout factors {0, 1/(1-p)} of the INPUT (F.dropout in front of lin2 / lin2_regr,
 * kernel/sgcn_img_snp.py:289,299), applied inside the kernels: y = (x * keep) W^T + b. */
int igcn_small_linear_fwd(int64_t R, int K, int C, const float* x, const float* keep, const float* W, const float* b,
                          float* y, void* stream);
int igcn_small_linear_bwd(int64_t R, int K, int C, const float* x, const float* keep, const float* W,
                          const float* dy, float* dx, float* dwb, float* scratch, void* stream);
/* Two such layers over inputs of the same [R, K] shape in one launch each way (lin2 and lin2_regr of the two heads):
 * per layer the arguments of igcn_small_linear_{fwd,bwd}. */
int igcn_small_linear_pair_fwd(int64_t R, int K, int C0, const float* x0, const float* keep0, const float* W0,
                               const float* b0, float* y0, int C1, const float* x1, const float* keep1,
                               const float* W1, const float* b1, float* y1, void* stream);
int igcn_small_linear_pair_bwd(int64_t R, int K, int C0, const float* x0, const float* keep0, const float* W0,
                               const float* dy0, float* dx0, float* dwb0, float* scratch0, int C1, const float* x1,
                               const float* keep1, const float* W1, const float* dy1, float* dx1, float* dwb1,
                               float* scratch1, void* stream);

/* SNP importance mask of cal_probability (kernel/sgcn_img_snp.py:147-151): out [B,S] = snps * sigmoid(p),
 * sp [S] = sigmoid(p).  Backward: dp [S] from dout [B,S] and/or dsp [S] (either may be NULL); snps gets no gradient. */
int igcn_snps_mask_fwd(int B, int S, const float* snps, const float* p, float* out, float* sp,
                       float* plain /* [B,S] copy of snps for the stacked batch, or NULL */, void* stream);
int igcn_snps_mask_bwd(int B, int S, const float* snps, const float* p, const float* dout, const float* dsp,
                       float* dp, void* stream);

/* Inputs of the two MLP heads (kernel/sgcn_img_snp.py:284-297) in one pass, R = passes * bsz rows (pass-major):
 *   out_z [R,W] = (img + cross) / 2;  out_lin [R,W+L] = out_z | latent;
 *   feat [R,W+L+P] = out_lin | (x * prob)[row % bsz]   (x [bsz,P] = data.x per sample, prob [P]; P = 0: feat NULL).
 * Backward: any of d_out_z / d_out_lin / d_feat may be NULL; d_mid [R,W] is the gradient of img and of cross,
 * d_latent [R,L], dx [bsz,P], dprob [P] (the last two only when P > 0).  W, L, P even. */
int igcn_head_inputs_fwd(int64_t R, int bsz, int W, int L, int P, const float* img, const float* cross,
                         const float* latent, const float* x, const float* prob, float* out_z, float* out_lin,
                         float* feat, void* stream);
int igcn_head_inputs_bwd(int64_t R, int bsz, int W, int L, int P, const float* d_out_z, const float* d_out_lin,
                         const float* d_feat, const float* x, const float* prob, float* d_mid, float* d_latent,
                         float* dx, float* dprob, void* stream);
/* The same backward that also takes the ReLU backward and the bias gradient of the layer whose POST-ReLU output `cross`
 * [R, W] is (relu(out_proj(.)), kernel/sgcn_img_snp.py:241-242; W = positions x D features): d_cross [R, W] = the gradient
 * of that layer's pre-activation (d_mid where cross > 0), db [D] = its column sums by feature, through the partial rows
 * db_part [igcn_head_inputs_bwd_blocks(R, W, L)][D] and a final reduction that is deferred while the stream defers.
 * D a power of two, 2 <= D <= 64, W % D == 0. */
/* igcn_head_inputs_fwd with the layer in front computed on the way: cross [R, W] = relu(o Wp^T + bp) per graph node
 * (o [R, W] = [R, W / D nodes, D features], Wp [D, D], bp [D]: relu(out_proj(.)), kernel/sgcn_img_snp.py:241-242), written
 * as well (the backward's ReLU mask, igcn_head_inputs_bwd_relu).  fp32 FMAs in k order.  D a power of two in [2, 64]
 * dividing W. */
int igcn_outproj_head_inputs_fwd(int64_t R, int bsz, int W, int L, int P, int D, const float* o, const float* Wp,
                                 const float* bp, const float* img, const float* latent, const float* x, const float* prob,
                                 float* cross, float* out_z, float* out_lin, float* feat, void* stream);
int igcn_head_inputs_bwd_blocks(int64_t R, int W, int L);
int igcn_head_inputs_bwd_relu(int64_t R, int bsz, int W, int L, int P, const float* d_out_z, const float* d_out_lin,
                              const float* d_feat, const float* x, const float* prob, float* d_mid, float* d_latent,
                              float* dx, float* dprob, const float* cross, float* d_cross, int D, float* db_part,
                              float* db, void* stream);

/* out[r, p*F + c] = parts[p][r, c], p < nparts <= 4: concatenation of the GCN layer outputs along the feature axis
 * (kernel/sgcn_img_snp.py:223-224, kernel/sgcn.py:376-377 `torch.cat(xs, dim=1)`).  F % 4 == 0, 16-byte aligned
 * tensors; `parts` is a HOST array of device pointers. */
int igcn_concat_cols(int64_t rows, int F, int nparts, const float* const* parts, float* out, void* stream);

/* Hand-over of a new batch to a captured train step (train.GraphedTrainStep.load; the reference's
 * `data = data.to(device)`, kernel/train_eval_sgcn_img_snps.py:517): up to IGCN_COPY_MULTI_MAX device-to-device copies
 * of arbitrary byte counts as ONE launch — x, edge_index, edge_attr, snps_feat, y, clini_score, tsne_fdim, clust_y, ptr,
 * edge_ptr of a batch are ten tensors of 1 KB - 1 MB, i.e. ten ~4 us copy launches in front of a 0.8 ms step otherwise.
 * dst / src / nbytes are HOST arrays [n]; ranges must not overlap.  16-byte lanes where both ends are aligned. */
#define IGCN_COPY_MULTI_MAX 16
int igcn_copy_multi(int n, void* const* dst, const void* const* src, const int64_t* nbytes, void* stream);

/* Dense image of a sparse map (the SNP <-> GO maps of kernel/go_model.py:208-215,281-282 as dense products, small
 * batches): image[c * image_stride + pos[k]] = val[c * val_stride + k] for c < channels, k < nnz (pos int64 [nnz]: row *
 * n_cols + col of every non-zero; the other entries of `image` are not touched), and the way back for the value
 * gradients: out [channels, nnz], out[c][k] = image[c * image_stride + pos[k]]. */
int igcn_image_put(int channels, int64_t nnz, const int64_t* pos, const float* val, int64_t val_stride, float* image,
                   int64_t image_stride, void* stream);
int igcn_image_take(int channels, int64_t nnz, const int64_t* pos, const float* image, int64_t image_stride, float* out,
                    void* stream);

/* Block-diagonal collation (Batch.from_data_list, batch.py:24-123) of B graphs drawn from a dataset of UNIFORM graphs held
 * as stacked device tensors, every key in ONE launch: idx int64 [B] (device) = the subjects; per key c < n (HOST arrays):
 * kind[c] == 0: dst_c [B, row] = src_c [idx[b], row] with row_bytes[c] bytes per graph (a multiple of 4);
 * kind[c] == 1: an `*index*` key — src_c [S, 2, E] int64, dst_c [2, B E] = concatenation along the last dim with graph b's
 * entries offset by b * nodes_per_graph (batch.py:98-104); row_bytes[c] = 2 * E * 8.  An idx[b] outside [0, n_subjects)
 * is not read: its rows are zeros, its index entries -1 (a consumer's plan build reports those in its status word). */
int igcn_gather_batch(int n, int B, int64_t nodes_per_graph, int64_t n_subjects, const int64_t* idx, void* const* dst,
                      const void* const* src, const int64_t* row_bytes, const int* kind, void* stream);

/* Measurement aid (bench.py roofline): the launch of igcn_gcn_propagate_fwd for (n_nodes, F) — thread =
 * (target, feature quad), or one wave per target when `dense` — with the body removed: mode 0 = empty kernel,
 * mode 1 = only the 16-byte store of each output row.  What the dispatch of that grid costs when no byte is read. */
int igcn_launch_floor(int64_t n_nodes, int F, int dense, int mode, float* out, void* stream);

/* Graph read-out of the `graph_pool=True` branch (kernel/sgcn_img_snp.py:230-235,246-252): PyG 2.0.2
 * global_mean_pool | global_max_pool | global_add_pool over the nodes of each graph, concatenated.  Uniform graphs of
 * `nodes_per_graph` nodes (the model's contract): x [n_graphs*nodes_per_graph, D] -> out [n_graphs, 3*D].
 * argmax [n_graphs, D] int32 (node index inside the graph, first maximum) is saved for the backward:
 * dx[g*R+r, d] = dout[g,d]/R + dout[g,2D+d] + (r == argmax[g,d]) * dout[g,D+d]. */
int igcn_graph_pool_fwd(int64_t n_graphs, int nodes_per_graph, int D, const float* x, float* out, int32_t* argmax,
                        void* stream);
int igcn_graph_pool_bwd(int64_t n_graphs, int nodes_per_graph, int D, const float* dout, const int32_t* argmax,
                        float* dx, void* stream);

/* Backward glue of out = act(A W^T + bias) (ops.Linear; lin1 / lin2 / GCNConv.lin of kernel/sgcn_img_snp.py:34-84):
 * g = dy * [y > 0] when `y` (the ReLU output) is given — then `g` receives the masked gradient the two GEMMs of
 * the backward read — and db[c] = sum_r g[r,c] (y == NULL: g unused, db = column sums of dy).  dy, y, g [rows, cols]
 * row-major, cols <= 256.  scratch: igcn_bias_grad_scratch_floats(rows, cols) floats. */
size_t igcn_bias_grad_scratch_floats(int64_t rows, int cols);
int igcn_bias_grad(int64_t rows, int cols, const float* dy, const float* y, float* g, float* db, float* scratch,
                   void* stream);
/* out[0:cols] = column sums of x [rows, cols]; out[cols : cols + zero_cols] = 0 (a gradient block known to vanish,
 * written by the launch that sums its neighbour: the key bias of nn.MultiheadAttention, kernel/sgcn_img_snp.py:240 —
 * a softmax over keys cannot see a key bias).  scratch: igcn_bias_grad_scratch_floats(rows, cols + zero_cols). */
int igcn_col_sums(int64_t rows, int cols, int zero_cols, const float* x, float* out, float* scratch, void* stream);
/* Two igcn_bias_grad / igcn_col_sums problems of the same [rows, cols] shape in one launch: per problem dy, optional
 * ReLU reference y (then g = dy * [y > 0] is written), db [cols + zero_cols] with zero_cols structurally zero
 * entries behind the sums, scratch of igcn_bias_grad_scratch_floats(rows, cols + zero_cols) floats. */
int igcn_bias_grad_pair(int64_t rows, int cols, const float* dy0, const float* y0, float* g0, float* db0,
                        int zero_cols0, float* scratch0, const float* dy1, const float* y1, float* g1, float* db1,
                        int zero_cols1, float* scratch1, void* stream);

/* `act`: 0 = none, 1 = ReLU; bit 0x100 = the output is a final parameter gradient (deferred reductions, below).
 * The split the library's launch heuristic prefers for (M, N, K): callers size `scratch` with it and pass it as
 * `split_k` (any other value >= 1 is honoured too). */
int igcn_gemm_f32_split_k(int64_t M, int64_t N, int64_t K);
int igcn_gemm_f32(int64_t M, int64_t N, int64_t K,
                  const float* A, int64_t sam, int64_t sak, const float* B, int64_t sbn, int64_t sbk,
                  const float* bias, float* C, int64_t ldc, int act, int split_k, float* scratch,
                  void* stream);

/* The same product with bf16 operands (BASELINE configs[4]: "bf16 feature transforms on CDNA4 MFMA"): A and B are
 * rounded to bf16 (round-to-nearest-even) while they are staged into LDS, multiplied on v_mfma_f32_16x16x32_bf16,
 * accumulated and written in fp32.  Tensors stay fp32 in memory: same arguments, launch shapes and split-K
 * convention as igcn_gemm_f32, so a model flag swaps the two (GCNConv.lin, the attention projections, lin1 /
 * lin1_regr of kernel/sgcn_img_snp.py:34-84,239-241).  Exact w.r.t. an fp64 product of the bf16-rounded operands up
 * to fp32 accumulation order. */
int igcn_gemm_bf16(int64_t M, int64_t N, int64_t K,
                   const float* A, int64_t sam, int64_t sak, const float* B, int64_t sbn, int64_t sbk,
                   const float* bias, float* C, int64_t ldc, int act, int split_k, float* scratch,
                   void* stream);
/* act | 0x200 (igcn_gemm_f32 / igcn_gemm_bf16; no bias, no activation, ldc == N): LEAVE the split-K slabs un-summed in
 * `scratch` ([igcn_gemm_effective_split(K, split_k)][M, N], slab order = K order) for a consumer that sums them while it
 * loads them (igcn_bn1d_fwd_slabs); with an effective split of 1 the product is written to C as usual. */
int igcn_gemm_effective_split(int64_t K, int split_k);

/* Batched-sum variant: C = sum_z A_z . B_z^T with A_z = A + z*a_batch, B_z = B + z*b_batch (element offsets),
 * every z covering the whole K; slabs are summed in z order.  Used for weight gradients whose reduction index is
 * (sample, node) over channel-major activations.  Long K is additionally sliced (<= 16 ways).
 * scratch: float[16*batch*M*N]. */
int igcn_gemm_f32_batched_sum(int64_t M, int64_t N, int64_t K, int batch,
                              const float* A, int64_t sam, int64_t sak, int64_t a_batch,
                              const float* B, int64_t sbn, int64_t sbk, int64_t b_batch,
                              float* C, int64_t ldc, float* scratch, void* stream);
/* `batch` independent products of one shape in one launch: C_z [M,N] = A_z B_z^T with A_z = A + z*a_batch,
 * B_z = B + z*b_batch, C_z = C + z*c_batch (element offsets).  split_k > 1 (see igcn_gemm_f32_split_k) needs
 * scratch of batch*split_k*M*N floats; the slabs of every product are summed by one more launch. */
int igcn_gemm_f32_batched(int64_t M, int64_t N, int64_t K, int batch, const float* A, int64_t sam, int64_t sak,
                          int64_t a_batch, const float* B, int64_t sbn, int64_t sbk, int64_t b_batch, float* C,
                          int64_t c_batch, int64_t ldc, int split_k, float* scratch, void* stream);
/* n (1..4) products of DIFFERENT shapes in one launch (the dX and dW products of a linear layer's backward): table
 * [n][16] int64 = {M, N, K, A, sam, sak, B, sbn, sbk, bias, C, ldc, act, split_k, scratch, bf16} per problem, each
 * field as the igcn_gemm_f32 argument of that name (pointers as integers); bf16 != 0 in the first problem: operands
 * rounded to bf16 as in igcn_gemm_bf16.  Same products as n igcn_gemm_f32 / igcn_gemm_bf16 calls; where K is split, the
 * slabs of a member are added in slab order (one launch for all members that need a plain sum) — the single-product entry
 * points choose their slab sum by shape (a tree for few outputs x many slabs, igcn_gemm_f32_batched likewise), so a split
 * product agrees with them to fp32 rounding, not bit for bit.  Within one entry point results are reproducible run to run. */
int igcn_gemm_f32_grouped(int n, const int64_t* table, void* stream);
/* Products QUEUED for `stream` (same table format; at most 4 waiting) and carried by the next igcn_gemm_f32_grouped on it
 * — if they fit beside its own products (4 in all) and share its operand type — as further members of that launch,
 * slab sums included.  For products that are ready together but issued by different parts of the host code (the heads'
 * first layers in the model's forward, the Gram matrices in the loss function).  igcn_gemm_rider_flush launches what
 * nobody carried.  Operands and outputs must stay alive until the carrying launch. */
int igcn_gemm_rider(void* stream, int n, const int64_t* table);
int igcn_gemm_rider_flush(void* stream);
/* Forget every rider still waiting on the stream (mask job, products) WITHOUT launching it — at the start of a step that
 * may follow one which failed between queueing a rider and its carrier (the buffers it points at may be gone). */
int igcn_rider_cancel(void* stream);
/* HOST-SIDE QUEUES OF THE LIBRARY — the complete list, with the ordering contract of each.  All are keyed by the stream
 * handle, hold raw device pointers that the CALLER keeps alive, never launch anything by themselves, and are empty at the
 * end of every step (igcn_stream_pending == 0; ig-gcn_amd/train.py asserts it under IGCN_DEBUG_SYNC=1):
 *   1. deferred reductions   igcn_reduce_defer(stream, 1) .. igcn_reduce_flush[_tick](stream): every "sum the block
 *      partials" launch issued on the stream in between (and igcn_reduce_rows_final, igcn_loss_final) is queued; the flush
 *      runs them in issue order in one launch.  The
 *      partial buffers must outlive the flush; igcn_reduce_defer(stream, 0) + flush also runs on the error path.
 *   2. the dropout rider     igcn_rider_dropout(stream, ...): at most ONE job; carried by the NEXT
 *      igcn_graph_plan_build_segmented[_rep], igcn_sgcn_front_fwd (which builds the plan itself) or igcn_dense_sgcn_fwd
 *      (complete graphs: no plan build; its first edge pass carries the job) on the stream and by nothing else;
 *      igcn_rider_flush launches it alone.
 *   3. product riders        igcn_gemm_rider(stream, ...): at most 4 products; carried, all or none, by the NEXT
 *      igcn_gemm_f32_grouped on the stream with room for them and the same operand type; igcn_gemm_rider_flush launches
 *      them alone.
 * A rider is only ever picked up by the entry point named above: any other launch on the stream passes it by, so the
 * caller issues rider -> (independent launches) -> carrier or flush and must not read the rider's outputs before that.
 * igcn_rider_cancel forgets 2 and 3 without launching (start of a step that may follow a failed one).
 * igcn_stream_pending(stream) = entries of 1 + jobs of 2 + products of 3 still waiting on that stream. */
int igcn_stream_pending(void* stream);

/* ------------------------------------------------------------------------------------------------
 * GO read-outs: per-node linear + BatchNorm1d(#nodes) + ReLU, fused — go_model.py:117-121,254
 * (conc_for_attention, D = dim_snps_atten), :123-128,255 (conc + B, D = 1), :130-136,278 (conc_D + B_D, D = 1):
 *   pre[b,n,:] = W x[b,:,n]   (x [B,F,N] channel-major, W [D,F])
 *   out[b,n,:] = relu((pre - mean_n)*rstd_n*gamma[n] + beta[n])      out [B,N,D]
 * BatchNorm1d(N) on [B,N,D] normalises every node over (batch, feature); training != 0 uses batch statistics
 * (biased variance) and updates running_mean/var (momentum, unbiased variance), else the running ones.
 * `groups` > 1 splits the B samples into equal consecutive groups that are normalised independently and update
 * the running statistics one after the other (= that many successive module calls; used to run the plain and the
 * masked forward pass of a train step as one launch).
 * scratch: igcn_node_linear_bn_scratch_floats(B,N).
 */
size_t igcn_node_linear_bn_scratch_floats(int B, int N, int groups);
int igcn_node_linear_bn_fwd(int B, int F, int N, int D, int groups, const float* x, const float* W,
                            const float* gamma, const float* beta, float* running_mean, float* running_var,
                            int training, float momentum, float eps,
                            const float* keep /*[B,N] dropout factors of the output, D == 1 only; NULL = none*/,
                            float* out, float* save_mean /*[groups,N]*/, float* save_rstd /*[groups,N]*/,
                            float* scratch, void* stream);
/* Outputs dx [B,F,N], dW [D,F], dgb [2,N] = (dgamma, dbeta).  dW = sum_{b,n} dpre[b,n,:] (x) x[b,:,n] is formed in
 * registers when D*F <= 16, else on the MFMA batched-sum GEMM.
 * scratch: igcn_node_linear_bn_bwd_scratch_floats(B,F,N,D,groups). */
size_t igcn_node_linear_bn_bwd_scratch_floats(int B, int F, int N, int D, int groups);
int igcn_node_linear_bn_bwd(int B, int F, int N, int D, int groups, int training, const float* x, const float* W,
                            const float* gamma, const float* beta, const float* save_mean, const float* save_rstd,
                            const float* dout, const float* keep, float* dx, float* dW, float* dgb, float* scratch,
                            void* stream);
/* Two read-outs of the SAME input x [B,F,N] in paired launches (go_model.py:254-255: conc_for_attention, D1 outputs per
 * node, and conc, D2 = 1 with fused dropout keep2): three launches forward and three backward for both, their grids
 * sharing the chip.  Per side the arguments of igcn_node_linear_bn_{fwd,bwd} (scratch sizes as there).
 * igcn_node_linear_bn_pair_supported(F, D1, D2) != 0 for the shapes the paired kernels cover (F = 5, D1 = 32, D2 = 1). */
int igcn_node_linear_bn_pair_supported(int F, int D1, int D2);
int igcn_node_linear_bn_pair_fwd(int B, int F, int N, int groups, const float* x, int training,
                                 int D1, const float* W1, const float* gamma1, const float* beta1,
                                 float* running_mean1, float* running_var1, float momentum1, float eps1, float* out1,
                                 float* save_mean1, float* save_rstd1, float* scratch1,
                                 int D2, const float* W2, const float* gamma2, const float* beta2,
                                 float* running_mean2, float* running_var2, float momentum2, float eps2,
                                 const float* keep2, float* out2, float* save_mean2, float* save_rstd2, float* scratch2,
                                 void* stream);
int igcn_node_linear_bn_pair_bwd(int B, int F, int N, int groups, int training, const float* x,
                                 int D1, const float* W1, const float* gamma1, const float* beta1,
                                 const float* save_mean1, const float* save_rstd1, const float* dout1, float* dx1,
                                 float* dW1, float* dgb1, float* scratch1,
                                 int D2, const float* W2, const float* gamma2, const float* beta2,
                                 const float* save_mean2, const float* save_rstd2, const float* dout2,
                                 const float* keep2, float* dx2, float* dW2, float* dgb2, float* scratch2, void* stream);

/* BatchNorm1d(C) (+ReLU when relu != 0) on a 2-D input [B,C] with the same grouped-statistics semantics —
 * the latent MLP of go_model.py:138-146.  save_mean/save_rstd are [groups,C]. */
int igcn_bn1d_fwd(int B, int C, int groups, const float* x, const float* gamma, const float* beta,
                  float* running_mean, float* running_var, int training, float momentum, float eps, int relu,
                  const float* keep /*[B,C] dropout factors of the output or NULL*/,
                  float* y, float* save_mean, float* save_rstd, void* stream);
int igcn_bn1d_bwd(int B, int C, int groups, int training, int relu, const float* x, const float* gamma,
                  const float* beta, const float* save_mean, const float* save_rstd, const float* dy,
                  const float* keep, float* dx, float* dgamma, float* dbeta, void* stream);
/* The same forward on a column whose split-K slabs the product in front left un-summed (igcn_gemm_f32 with act | 0x200):
 * the latent MLP's wide layer, kernel/go_model.py:138-146 (Linear -> BatchNorm1d -> ReLU -> Dropout).  `slabs`
 * [n_slabs][B, C] are summed in slab order while the column is loaded; x_out [B, C] receives the sum (igcn_bn1d_bwd's
 * operand).  groups <= 2 and B / groups <= 1024 (igcn_bn1d_fwd_supported). */
int igcn_bn1d_fwd_supported(int B, int groups);
int igcn_bn1d_fwd_slabs(int B, int C, int groups, const float* slabs, int n_slabs, float* x_out, const float* gamma,
                        const float* beta, float* running_mean, float* running_var, int training, float momentum,
                        float eps, int relu, const float* keep, float* y, float* save_mean, float* save_rstd,
                        void* stream);

/* Dropout factors of every site of a forward pass in ONE launch (the reference draws them site by site: nn.Dropout /
 * nn.Dropout2d / F.dropout, kernel/go_model.py:104,113,128,136,143, kernel/sgcn_img_snp.py:289,299).  out [total]:
 * element i of segment k (segments given by their END offsets, HOST arrays, <= igcn_dropout_max_segments()) is 0 with probability seg_p[k],
 * else 1/(1-seg_p[k]); consumed as `keep` by igcn_nodes_ln_*, igcn_node_linear_bn_*, igcn_bn1d_*, igcn_small_linear_*.
 * Counter-based integer-hash generator; `state` = device uint64[igcn_dropout_state_words()], word 0 = the stream
 * counter (seed), every other word 0 (arrival counts, left at 0 by every launch): the last workgroup of a launch
 * advances the counter, so every replay of a captured launch draws fresh masks. */
int igcn_dropout_max_segments(void);   /* sites one igcn_dropout_masks launch takes; callers split longer lists */
int igcn_dropout_state_words(void);
int igcn_dropout_masks(int64_t total, int n_segments, const int64_t* seg_end, const float* seg_p, void* state,
                       float* out, int n_counters, int64_t* const* counters, int64_t counter_inc, void* stream);
/* The same job as a RIDER: not launched, but queued for `stream` (at most one per stream) and carried by the next
 * igcn_graph_plan_build_segmented[_rep] on that stream — extra workgroups of the plan build's grid draw the masks (both
 * launches depend on nothing a train step computes; as two roles of one grid they overlap).  igcn_rider_flush launches a
 * job nobody carried as a launch of its own (nothing waiting: nothing happens). */
int igcn_rider_dropout(void* stream, int64_t total, int n_segments, const int64_t* seg_end, const float* seg_p, void* state,
                       float* out, int n_counters, int64_t* const* counters, int64_t counter_inc);
int igcn_rider_flush(void* stream);
/* counters (HOST array of <= 8 device int64 pointers, or n_counters = 0): each is bumped by counter_inc by the launch —
 * BatchNorm's num_batches_tracked (kernel/go_model.py:119-146), which advance exactly when the masks are drawn.
 *
 * out[i] = sum_k parts[k][i], k < n <= 4 (HOST array of device pointers, 16-byte aligned): the gradient of a tensor
 * with several consumers in one launch instead of autograd's pairwise adds (ops.GradFan). */
int igcn_sum_n(int64_t numel, int n, const float* const* parts, float* out, void* stream);
/* The same sum for a LEAF's gradient (nothing reads it before the optimiser): while igcn_reduce_defer is on it joins the
 * deferred final reductions and is performed by igcn_reduce_flush's single launch — parts must stay alive until then;
 * otherwise it is igcn_sum_n.  Same summation order either way. */
int igcn_sum_n_final(int64_t numel, int n, const float* const* parts, float* out, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Mask regulariser — loss_probability, kernel/sgcn_img_snp.py:153-181:
 *   loss = mean r_x(sigmoid(prob)) + mean r_e(e) + mean r_x(sigmoid(snps_prob)),
 *   r(p) = l1*|p| - ent*(p log(p+eps) + (1-p) log(1-p+eps))
 * prob [n_prob] and snps [n_snps] are logits, e [n_edge] the edge mask of igcn_edge_mask_fwd.
 * scratch: float[1024].  gout: device scalar d(loss_total)/d(loss).
 */
/* loss == NULL: no final sum — the igcn_mask_reg_blocks(n_prob + n_edge + n_snps) block partials stay in scratch for
 * the consumer to add up (igcn_loss_head_fwd, prob_rows). */
int igcn_mask_reg_blocks(int64_t n_total);
int igcn_mask_reg_fwd(int64_t n_prob, int64_t n_edge, int64_t n_snps, const float* prob, const float* e,
                      const float* snps, float l1_x, float ent_x, float l1_e, float ent_e, float eps,
                      float* loss, float* scratch, void* stream);
int igcn_mask_reg_bwd(int64_t n_prob, int64_t n_edge, int64_t n_snps, const float* prob, const float* e,
                      const float* snps, float l1_x, float ent_x, float l1_e, float ent_e, float eps,
                      const float* gout, float* dprob, float* de, float* dsnps, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Batch-level losses from ONE Gram matrix G = s s^T [B,B] of the fused features s [B, R*D]:
 *   out[0] = consist_loss (kernel/sgcn_img_snp.py:183-196) = sum_ij Lap_ij G_ij / B^2
 *   out[1] = OrthogonalConstraint (:198-205) = (sum_ij G_ij^2/(G_ii G_jj) - 2B + RD)/B^2
 * igcn_rbf_laplacian builds Lap = diag(W1) - W, W = exp(-gamma*||t_i-t_j||^2) (util/image_cluster.py:15-31;
 * t == NULL: W = 1).  The backward writes S = dG + dG^T [B,B] (ds = S s); gout [2] is a device array.
 * `groups` Gram matrices that share the Laplacian (the passes of a batched sweep: G [groups,B,B], out / gout
 * [groups,2], S [groups,B,B]) take one launch.  scratch (fwd): float[2 B groups].
 */
int igcn_rbf_laplacian(int B, int T, float gamma, const float* t, float* Lap, void* stream);
/* out == NULL: no final sum — the row partials [B][groups*2] stay in scratch (igcn_loss_head_fwd, gram_rows). */
int igcn_gram_loss_fwd(int B, int RD, int groups, const float* G, const float* Lap, float* out, float* scratch,
                       void* stream);
int igcn_gram_loss_bwd(int B, int groups, const float* G, const float* Lap, const float* gout, float* S,
                       void* stream);
/* igcn_gram_loss_fwd with the RBF Laplacian of consist_loss (util/image_cluster.py:15-31; tsne [B, T], or NULL: W = 1)
 * evaluated inside the row walk instead of read: lap_out [B, B] is an OUTPUT (igcn_gram_loss_bwd reads it) — no
 * igcn_rbf_laplacian launch in the train step. */
int igcn_gram_loss_fwd_rbf(int B, int RD, int groups, const float* G, const float* tsne, int T, float gamma, float* lap_out,
                           float* out, float* scratch, void* stream);
/* ... that also writes S [groups, B, B] = the output of igcn_gram_loss_bwd for the upstream gradient gout [groups, 2]
 * given here as a HOST array (groups <= 4), bit for bit: a train step knows its d loss / d (consist, orth) — the loss
 * weights — when it launches the forward, and then has no Gram-loss backward launch. */
int igcn_gram_loss_fwd_rbf_unit(int B, int RD, int groups, const float* G, const float* tsne, int T, float gamma,
                                float* lap_out, float* out, float* scratch, const float* gout, float* S, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Cross-attention core of nn.MultiheadAttention (kernel/sgcn_img_snp.py:240) on the projection outputs in place:
 *   q [B,Lq,D] (query projection), kv [B,Lk,2,D] (key|value projection as ONE GEMM output), D = H*head_dim
 *   o [B,Lq,D] = per head softmax(q k^T / sqrt(head_dim)) v, heads concatenated; lse [B,H,Lq] saved for the backward
 * One workgroup per (sample, head); no head transposes or contiguous copies on either side.  The backward returns
 * dq [B,Lq,D] and dkv [B,Lk,2,D] in the layouts the projection-gradient GEMMs read.
 * head_dim <= 32 runs on the matrix cores (exact-fp32 MFMA, any Lq, head columns zero-padded to a multiple of 4 in
 * LDS); when K, V — and Q, dO in the backward — of one head fit LDS one workgroup per (sample, head) keeps them
 * resident, otherwise the other side of the attention is streamed through LDS in chunks (512 queries x 1300 keys of
 * the 512-ROI configuration).  A VALU kernel (head_dim in {4,..,24} step 4, Lq <= 256) remains for A/B runs.
 * igcn_attn_core_lds_bytes: dynamic LDS needed, 0 = shape not covered.
 */
size_t igcn_attn_core_lds_bytes(int D, int H, int Lq, int Lk, int backward);
size_t igcn_attn_core_bwd_scratch_floats(int B, int H, int Lq);
int igcn_attn_core_fwd(int B, int D, int H, int Lq, int Lk, const float* q, const float* kv, float* o, float* lse,
                       void* stream);
int igcn_attn_core_bwd(int B, int D, int H, int Lq, int Lk, const float* q, const float* kv, const float* o,
                       const float* lse, const float* dout, float* dq, float* dkv,
                       float* scratch /* igcn_attn_core_bwd_scratch_floats(): rowsum(o*do) of the chunked path */,
                       void* stream);
/* The same core with bf16 OPERANDS on v_mfma_f32_16x16x32_bf16 (BASELINE configs[4] "bf16 feature transforms on
 * CDNA4 MFMA"): q k^T, p v and the three gradient products take operands rounded to bf16 and accumulate in fp32; the
 * softmax, lse and delta stay fp32, and every tensor stays fp32 in HBM (same layouts and signatures as above; scratch
 * is always needed).  head_dim must be 16 (igcn_attn_core_bf16_supported); any Lq, Lk (both sides streamed through
 * LDS in chunks).  Tolerance vs the fp32 core: ~4e-3 relative on o, ~1e-2 on the gradients (tests/test_gpu_ops.py). */
int igcn_attn_core_bf16_supported(int D, int H, int Lq, int Lk);
int igcn_attn_core_bf16_fwd(int B, int D, int H, int Lq, int Lk, const float* q, const float* kv, float* o, float* lse,
                            void* stream);
int igcn_attn_core_bf16_bwd(int B, int D, int H, int Lq, int Lk, const float* q, const float* kv, const float* o,
                            const float* lse, const float* dout, float* dq, float* dkv, float* scratch, void* stream);

/* The attention core on SPLIT bf16 operands (csrc/attn_split.hip; head_dim 16): every fp32 operand is carried as a bf16
 * head and a bf16 remainder (x = hi + lo to 2^-17 |x|) and every product runs on v_mfma_f32_16x16x32_bf16 — fp32-grade
 * results (the 1e-4 / 1e-3 bounds of igcn_attn_core_fwd / _bwd hold, tests/test_gpu_ops.py) at the bf16 rate of the
 * matrix cores.  Arguments as igcn_attn_core_fwd / igcn_attn_core_bwd. */
int igcn_attn_core_split_supported(int D, int H, int Lq, int Lk);
int igcn_attn_core_split_fwd(int B, int D, int H, int Lq, int Lk, const float* q, const float* kv, float* o, float* lse,
                             void* stream);
/* (backward: one workgroup per (sample, head), <= 128 queries and >= 128 keys — igcn_attn_core_split_bwd_supported) */
int igcn_attn_core_split_bwd_supported(int D, int H, int Lq, int Lk);
int igcn_attn_core_split_bwd(int B, int D, int H, int Lq, int Lk, const float* q, const float* kv, const float* o,
                             const float* lse, const float* dout, float* dq, float* dkv, float* scratch, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Sparse SNP<->GO maps with learnable non-zeros — gene encoding go_model.py:208-215 (C=2 channels,
 * rows = GO nodes, cols = SNPs) and gene decoding :281-282 (C=1, rows = SNPs, cols = GO nodes):
 *   y[b,c,i] = sum_{k in row i} val[c,k] * x[b, col[k]]          x [B,J], y [B,C,I], val [C,nnz]
 * Backward needs the transposed structure: t_ptr [J+1], t_row [nnz] (row of each entry in column order),
 * t_k [nnz] (its position in the row-major value array).
 * The structure is shared by all samples and small: the kernels read it once per workgroup and reuse it across a tile
 * of samples whose operand rows sit in LDS (short lists: thread per output row; ~150-entry lists: workgroup per sample
 * with its vector in LDS, wave per output; value gradients: a few samples' vectors in LDS, thread per non-zero).
 * dval is a parameter gradient: its final sum over the workgroups' partial rows takes part in the deferred
 * reductions (igcn_reduce_defer).  scratch: igcn_spmm_bwd_scratch_floats(...) floats.
 */
int igcn_spmm_fwd(int B, int C, int I, int J, int64_t nnz, const int32_t* row_ptr, const int32_t* col,
                  const float* val, const float* x, float* y, void* stream);
size_t igcn_spmm_bwd_scratch_floats(int B, int C, int I, int J, int64_t nnz);
int igcn_spmm_bwd(int B, int C, int I, int J, int64_t nnz, const int32_t* row_ptr, const int32_t* col,
                  const int32_t* row_of /*[nnz]*/, const int32_t* t_ptr, const int32_t* t_row, const int32_t* t_k,
                  const float* val, const float* x, const float* dy,
                  float* dx /*[B,J] or NULL*/, float* dval /*[C,nnz] or NULL*/,
                  float* scratch /* needed when dval != NULL */, void* stream);
/* The value rows of the C channels at a constant stride instead of one [C, nnz] tensor: val + c * val_stride (floats,
 * >= nnz) — the per-channel parameter vectors of go_model.py:67 as the optimiser lays them out in its flat buffer, so
 * that no concatenation runs in front of every step.  dval stays one contiguous [C, nnz] output. */
int igcn_spmm_fwd_strided(int B, int C, int I, int J, int64_t nnz, const int32_t* row_ptr, const int32_t* col,
                          const float* val, int64_t val_stride, const float* x, float* y, void* stream);
int igcn_spmm_bwd_strided(int B, int C, int I, int J, int64_t nnz, const int32_t* row_ptr, const int32_t* col,
                          const int32_t* row_of, const int32_t* t_ptr, const int32_t* t_row, const int32_t* t_k,
                          const float* val, int64_t val_stride, const float* x, const float* dy, float* dx, float* dval,
                          float* scratch, void* stream);
/* The value-gradient half of igcn_spmm_bwd for one or two maps in ONE launch (call igcn_spmm_bwd with dval = NULL for
 * the input gradients): table [n][12] int64 = {B, C, I, J, nnz, col, row_of, x, dy, dval, scratch, 0} per map. */
int igcn_spmm_bwd_dval_multi(int n, const int64_t* table, void* stream);

/* ------------------------------------------------------------------------------------------------
 * GO attention-GCN encoder layer, all samples at once — replaces the dense transforms, the edge gathers
 * and the PER-SAMPLE python loop of go_model.py:226-244 (helper :182-186, attention_adj :173-180, gcn :170):
 *   x_in = W_inc x, x_s = W_s x                              (per node, fin -> fout)
 *   s_e  = exp(tanh(a_in[:fout].x_in[row_e] + a_in[fout:].x_in[col_e]))
 *   y[b,:,r] = sum_{e in row r} (s_e / sum_{e' in row r} s_e') x_in[col_e] + x_s[r] * sigmoid(a_s.x_s[r])
 * x [B,fin,N] and y [B,fout,N] channel-major.  Supported (fin,fout): (2,5) (5,5).
 */
int igcn_go_attn_fwd(int B, int N, int fin, int fout, const int32_t* row_ptr, const int32_t* col,
                     const float* x, const float* w_inc, const float* w_s, const float* a_in, const float* a_s,
                     float* y, void* stream);
/* Backward in O(nnz*f) (the reference's autograd forms a dense N x N product per sample and layer).
 * t_ptr/t_row: the transposed structure (for each column the rows that read it).
 * Outputs: dx [B,fin,N]; dparams float[ 2*fout*fin + 2*fout + fout ] = (dW_inc, dW_s, da_in, da_s).
 * Samples whose operands fit a CU's LDS ((fin+fout+4)*N*4 <= 160 KB, N <= 4096) run in an LDS-resident kernel, one
 * workgroup per sample; larger ones in global-memory kernels (hub columns walked by the whole wave).
 * scratch floats: igcn_go_attn_bwd_scratch_floats(B,N,fin,fout).  dparams is a FINAL reduction of block partials kept
 * in scratch: while igcn_reduce_defer is on it is performed by igcn_reduce_flush (scratch must stay alive until then). */
size_t igcn_go_attn_bwd_scratch_floats(int B, int N, int fin, int fout);
/* `walk_order` (device int32[igcn_go_attn_walk_slots(N, fin, fout)], or NULL = node order): the thread -> node map of
 * the LDS-resident kernel's column walks, from igcn_go_attn_walk_order(N, fin, fout, t_ptr on the HOST, order out on
 * the HOST) — structure-only, computed once per hierarchy and layer shape: nodes sorted by their number of readers and
 * dealt to the waves so that lanes walk lists of equal length and waves carry equal totals.  Results do not depend on
 * it.  igcn_go_attn_bwd_threads: the workgroup size the LDS-resident kernel runs with for that shape (512 when two
 * workgroups then fit a CU, else 1024) — the slot count is its multiple. */
int igcn_go_attn_bwd_threads(int N, int fin, int fout);
int igcn_go_attn_walk_slots(int N, int fin, int fout);
int igcn_go_attn_walk_order(int N, int fin, int fout, const int32_t* t_ptr_host, int32_t* order_host);
int igcn_go_attn_bwd(int B, int N, int fin, int fout, const int32_t* row_ptr, const int32_t* col,
                     const int32_t* t_ptr, const int32_t* t_row, const int32_t* walk_order,
                     const float* x, const float* w_inc, const float* w_s, const float* a_in, const float* a_s,
                     const float* dy, float* dx, float* dparams, float* scratch, void* stream);

/* ------------------------------------------------------------------------------------------------
 * LayerNorm over the NODE axis + ReLU + node dropout + hierarchical pooling — go_model.py:246-251
 * (encoder) and :273-275 (decoder): for every (sample, channel) row of y [B,f,N]
 *   z = relu( (y-mean)/sqrt(var+eps) * gamma[n] + beta[n] ) * keep[b,n]      (keep may be NULL)
 * and only nodes n >= pool are written: z [B,f,N-pool].  Saves mean, rstd [B*f].
 */
int igcn_nodes_ln_fwd(int B, int f, int N, int pool, float eps, const float* y, const float* gamma,
                      const float* beta, const float* keep, float* z, float* mean, float* rstd, void* stream);
/* Outputs dy [B,f,N], dgb [2,N] (dgamma row then dbeta row).
 * scratch: igcn_nodes_ln_bwd_scratch_floats(B,f,N) floats. */
size_t igcn_nodes_ln_bwd_scratch_floats(int B, int f, int N);
int igcn_nodes_ln_bwd(int B, int f, int N, int pool, const float* y, const float* gamma, const float* beta,
                      const float* keep, const float* mean, const float* rstd, const float* dz,
                      float* dy, float* dgb, float* scratch, void* stream);
/* The same backward in two calls: the input gradient now, the affine gradients (d gamma | d beta: parameter gradients,
 * read by nobody before the optimiser) of up to four layers later, in ONE launch.  table [n][13] int64 =
 * {B, f, N, pool, y, gamma, beta, keep, mean, rstd, dz, scratch, dgb} per layer (pointers as integers). */
int igcn_nodes_ln_bwd_dy(int B, int f, int N, int pool, const float* y, const float* gamma, const float* beta,
                         const float* keep, const float* mean, const float* rstd, const float* dz, float* dy,
                         void* stream);
int igcn_nodes_ln_bwd_affine_multi(int n, const int64_t* table, void* stream);

/* A GO layer's backward WITH the backward of the LayerNorm block behind it (go_model.py:219-251 encoder,
 * :262-275 decoder: y = layer(x); z = dropout(relu(LN(y)))[.., pool:]).  The layer's LDS-resident backward holds one
 * sample per workgroup, so it forms d loss / d y from (y, dz, gamma, beta, keep, mean, rstd) while copying it in — the
 * LayerNorm's own dX launch and the d y tensor disappear — and leaves d gamma | d beta summed over the sample's channels
 * in part [B][2][N], reduced over samples by the (deferred) final reduction into dgb [2, N].  igcn_go_*_ln_fused_ok:
 * 1 when the sizes qualify (LDS-resident layer, N and pool multiples of 4, N / 4 <= workgroup size); otherwise use
 * igcn_nodes_ln_bwd* followed by igcn_go_attn_bwd / igcn_go_decode_bwd.  Every tensor 16-byte aligned.  dx, dparams and
 * scratch as the plain entry points; part: igcn_go_ln_part_floats(B, N) floats, alive until the reductions ran.
 * igcn_go_attn_ln_bwd's dz2 / dz3: when z feeds several consumers (the encoder output: two read-outs and the decoder) their
 * gradients are added while they are loaded, in that order — the sum igcn_sum_n would have written. */
int igcn_go_attn_ln_fused_ok(int N, int fin, int fout, int pool);
int igcn_go_decode_ln_fused_ok(int Nin, int Nout, int fin, int fout);
size_t igcn_go_ln_part_floats(int B, int N);
int igcn_go_attn_ln_bwd(int B, int N, int fin, int fout, const int32_t* row_ptr, const int32_t* col,
                        const int32_t* t_ptr, const int32_t* t_row, const int32_t* walk_order,
                        const float* x, const float* w_inc, const float* w_s, const float* a_in, const float* a_s,
                        int pool, const float* y, const float* gamma, const float* beta, const float* keep,
                        const float* mean, const float* rstd, const float* dz, const float* dz2 /*or NULL*/,
                        const float* dz3 /*or NULL*/, float* dx, float* dparams, float* dgb, float* scratch, float* part,
                        void* stream);
int igcn_go_decode_ln_bwd(int B, int Nin, int Nout, int fin, int fout, const int32_t* row_ptr,
                          const int32_t* t_ptr, const int32_t* t_row, const float* x, const float* w_out,
                          const float* w_sout, const float* y, const float* gamma, const float* beta,
                          const float* keep, const float* mean, const float* rstd, const float* dz, float* dx,
                          float* dparams, float* dgb, float* scratch, float* part, void* stream);

/* ------------------------------------------------------------------------------------------------
 * GO decoder layer (mean aggregation down the hierarchy) — go_model.py:262-272, batch_mul :197-201:
 *   y[b,:,r] = (1/deg_r) sum_{e in row r} W_out x[b,:,col_e]  +  (r >= off ? W_sout x[b,:,r-off] : 0)
 * x [B,fin,Nin], y [B,fout,Nout], off = Nout-Nin.  Supported (fin,fout): (5,5) (5,2).
 */
int igcn_go_decode_fwd(int B, int Nin, int Nout, int fin, int fout, const int32_t* row_ptr, const int32_t* col,
                       const float* x, const float* w_out, const float* w_sout, float* y, void* stream);
/* Outputs dx [B,fin,Nin], dparams float[2*fout*fin] = (dW_out, dW_sout). */
size_t igcn_go_decode_bwd_scratch_floats(int B, int Nin, int fin, int fout);
int igcn_go_decode_bwd(int B, int Nin, int Nout, int fin, int fout, const int32_t* row_ptr,
                       const int32_t* t_ptr, const int32_t* t_row,
                       const float* x, const float* w_out, const float* w_sout, const float* dy,
                       float* dx, float* dparams, float* scratch, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Adam step over a flat fp32 parameter buffer — torch.optim.Adam(lr, weight_decay=0) at
 * kernel/train_eval_sgcn_img_snps.py:108,547.  `step` is a device int32 that the kernel's caller
 * increments (igcn_adam_step does it: one extra 1-thread kernel) so the call is graph-capturable.
 * `lr` is a DEVICE float[1] read by the kernel at run time: the reference's schedule
 * (`param_group['lr'] = lr_decay_factor * param_group['lr']`, kernel/train_eval_sgcn_img_snps.py:169-171) writes that
 * scalar, and a launch captured into a hipGraph sees the new rate at its next replay.
 */
int igcn_adam_step(int64_t n, float* param, const float* grad, float* exp_avg, float* exp_avg_sq,
                   int32_t* step, const float* lr, float beta1, float beta2, float eps, float grad_scale,
                   void* stream);

/* Multi-tensor forms: `table` is a device array int64[n_tensors][4] = {param, grad, exp_avg, exp_avg_sq}
 * pointers, `numel` int64[n_tensors].  A zero grad pointer skips the tensor (torch's Adam skips parameters whose
 * .grad is None).  igcn_pack_grads copies the gradients (zeros where missing) to flat[offset[t]...] — the bucket of
 * the data-parallel all-reduce. */
int igcn_adam_step_multi(int n_tensors, const int64_t* table, const int64_t* numel, int32_t* step,
                         const float* lr, float beta1, float beta2, float eps, float grad_scale, void* stream);
/* ..._ticked: `step` has been advanced already — by igcn_reduce_flush_tick, the launch that ends the backward — so
 * no one-thread counter launch goes in front. */
int igcn_adam_step_ticked(int64_t n, float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int32_t* step,
                          const float* lr, float beta1, float beta2, float eps, float grad_scale, void* stream);
int igcn_adam_step_multi_ticked(int n_tensors, const int64_t* table, const int64_t* numel, int32_t* step,
                                const float* lr, float beta1, float beta2, float eps, float grad_scale, void* stream);
/* igcn_adam_step_multi over a precomputed block list (the shapes of a model do not change between steps): block b covers
 * elements [blk_off[b], blk_off[b] + igcn_adam_chunk()) of tensor blk_tensor[b] — one workgroup per block instead of a
 * (96, n_tensors) grid whose workgroups mostly find nothing to do.  ticked != 0: `step` has been advanced already. */
int igcn_adam_chunk(void);
int igcn_adam_step_blocks(int n_blocks, const int64_t* table, const int64_t* numel, const int32_t* blk_tensor,
                          const int32_t* blk_off, int32_t* step, const float* lr, float beta1, float beta2, float eps,
                          float grad_scale, int ticked, void* stream);
int igcn_pack_grads(int n_tensors, const int64_t* table, const int64_t* numel, const int64_t* offset,
                    float* flat, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Deferred reductions of a backward pass.  About thirty kernels of the train step's backward end in a small "sum the
 * block partials" launch whose output is a parameter gradient that nothing reads before the optimiser (bias and
 * LayerNorm / BatchNorm affine gradients, split-K weight-gradient slabs, ...).  After igcn_reduce_defer(stream, 1)
 * those launches ON THAT STREAM are QUEUED instead (one queue per stream: trainers on different streams do not see each
 * other's partials; the partial buffers — the `scratch` arguments of the calls — must then stay alive until the
 * flush), and igcn_reduce_flush(stream) performs the stream's queued reductions in ONE launch, with the arithmetic
 * and summation order of the stand-alone kernels.  igcn_gemm_f32 / igcn_gemm_bf16 take part when bit 0x100 of `act`
 * marks their output as such a gradient.  Host-side state only: capturable like any other launch.
 * igcn_reduce_pending: queued entries over all streams. */
int igcn_reduce_defer(void* stream, int on);
int igcn_reduce_pending(void);
int igcn_reduce_flush(void* stream);
/* the flush + `*step_counter += 1` (device int32: the optimiser's step) by the same launch (a one-thread launch of its
 * own when nothing is queued) */
int igcn_reduce_flush_tick(void* stream, int32_t* step_counter);
/* out[j] = sum_r partial[r * ld + j] (j < n) as such a final reduction: queued while the stream defers, a launch of its own
 * otherwise — for the partial rows a kernel left behind for a parameter gradient (igcn_head_loss_fwd's wpart). */
int igcn_reduce_rows_final(const float* partial, int64_t rows, int64_t ld, int n, float* out, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Gradient exchange of the data-parallel step (the reference has no multi-GPU code): one RCCL
 * communicator per process (one process per GPU), ONE all-reduce(sum) of the flat fp32 gradient bucket that
 * igcn_pack_grads fills, in place, enqueued on `stream` — i.e. on the launch stream between the pack kernel and
 * igcn_adam_step(grad_scale = 1/world_size), capturable into the step's hipGraph.  RCCL is resolved at run time
 * (the librccl already loaded in the process — PyTorch-ROCm's — else dlopen): IGCN_ERR_UNSUPPORTED when absent.
 *   rank 0: igcn_comm_get_unique_id(buf[igcn_comm_unique_id_bytes()]) -> host-side broadcast of the bytes (any
 *   channel: torch.distributed store, MPI, a file) -> every rank: igcn_comm_init(W, rank, bytes, &comm) with its
 *   GPU current.  These three calls are host-blocking set-up, not stream work; igcn_comm_allreduce never blocks. */
int igcn_comm_unique_id_bytes(void);
int igcn_comm_get_unique_id(void* out_bytes);
int igcn_comm_init(int world_size, int rank, const void* unique_id, void** comm_out);
int igcn_comm_allreduce(void* comm, float* buf, int64_t n, void* stream);
int igcn_comm_destroy(void* comm);

/* ------------------------------------------------------------------------------------------------
 * Loss head of train() — kernel/train_eval_sgcn_img_snps.py:525-543 — on the STACKED outputs of the batched sweep
 * (rows [0,B) = plain pass :521, rows [B,2B) = isExplain pass :523):
 *   terms[7] = {lam0*nll(logp[:B],y), lam0*nll(logp[B:],y), lam1*(mse_1+mse_2)/2, lam2*prob,
 *               lam3*(sse_1+sse_2)/2 (snps reconstruction), lam4*(consist_1+consist_2)/2, lam5*orth_1}
 *   loss = hp_ce*terms[0] + hp_mi*terms[1] + terms[2] + ... + terms[6]
 * logp [2B,C] log-probabilities, y int64 [B], reg [2B,NR], clin [B*NR], x_hat [2B,S], snps [B,S],
 * gram [2][2] = igcn_gram_loss_fwd outputs of the two passes, prob [1] = igcn_mask_reg_fwd output.
 * lam6 is a HOST array (read at call time).  Backward: gout [1] device scalar; dgram [4], dprob [1].
 * from_logits != 0: `logp` holds the raw class scores [2B,C]; F.log_softmax (kernel/sgcn_img_snp.py:305) is taken
 * inside and written to logp_out [2B,C]; the backward, given that logp, returns the gradient of the raw scores.
 * gram_rows / prob_rows > 1: gram [gram_rows][4] / prob [prob_rows] are the un-reduced partials of
 * igcn_gram_loss_fwd (out == NULL) / igcn_mask_reg_fwd (loss == NULL); their rows are summed here (the gradient of
 * every row is dgram / dprob).
 */
int igcn_loss_head_fwd(int B, int C, int NR, int S, const float* logp, int from_logits, float* logp_out,
                       const int64_t* y, const float* reg, const float* clin, const float* x_hat, const float* snps,
                       const float* gram, int gram_rows, const float* prob, int prob_rows, const float* lam6,
                       float hp_ce, float hp_mi, float* loss, float* terms, void* stream);
int igcn_loss_head_bwd(int B, int C, int NR, int S, const int64_t* y, const float* reg, const float* clin,
                       const float* x_hat, const float* snps, const float* logp, const float* lam6, float hp_ce,
                       float hp_mi, const float* gout, float* dlogp, float* dreg, float* dxhat, float* dgram,
                       float* dprob, void* stream);

/* THE OUTPUT HEADS AND THE LOSS HEAD OF A TRAIN STEP AS ONE LAUNCH (kernel/sgcn_img_snp.py:289-290,300-301,305 lin2 /
 * lin2_regr / log_softmax on the stacked sweep; train() :525-530 the cross-entropy, regression and reconstruction terms)
 * together with their BACKWARD for an upstream gradient of one: x1 / x2 [2B, K] = the heads' hidden features (keep1 / keep2
 * [2B, K]: dropout factors of those inputs, or NULL), W1 [C, K] + b1 = lin2, W2 [NR, K] + b2 = lin2_regr, y [B] int64, clin
 * [B, NR], x_hat [2B, S], snps [B, S]; K / 4 a power of two <= 64, C, NR <= 4 (igcn_head_loss_supported).  Outputs: logp_out
 * [2B, C] = log_softmax of the scores, reg_out [2B, NR]; dx1 / dx2 [2B, K] = d loss / d (x1, x2), dxhat [2B, S]; wpart
 * [blocks, C K + C + NR K + NR] = per-workgroup partial rows of (dW1 | db1 | dW2 | db2) for igcn_reduce_rows_final; dgram [4],
 * dprob [1] = d loss / d (Gram terms, regulariser); parts [blocks, 4] = partial sums of the four row-wise loss terms
 * (blocks = igcn_head_loss_blocks(B, K)).  Same arithmetic as igcn_small_linear_pair_fwd -> igcn_loss_head_fwd_grads ->
 * igcn_small_linear_pair_bwd.
 * igcn_loss_final: the loss VALUE from the partial sums of this launch, of igcn_gram_loss_fwd (gram [gram_rows, 4]) and of the
 * mask regulariser (prob [prob_rows]); wts [10] DEVICE floats = {lam[0..5], hp_ce, hp_mi, B, NR}; out [8] = loss, terms[7].
 * Nothing of the backward depends on it: while the stream defers its reductions (igcn_reduce_defer) it joins the flush as
 * one more workgroup, otherwise it is a launch of its own. */
int igcn_head_loss_supported(int K, int C, int NR);
int igcn_head_loss_blocks(int B, int K);
int igcn_head_loss_fwd(int B, int K, int C, int NR, int S, const float* x1, const float* keep1, const float* W1,
                       const float* b1, const float* x2, const float* keep2, const float* W2, const float* b2,
                       const int64_t* y, const float* clin, const float* x_hat, const float* snps,
                       const float* lam6 /*HOST [6]*/, float hp_ce, float hp_mi, float* logp_out, float* reg_out, float* dx1,
                       float* dx2, float* dxhat, float* parts, float* wpart, float* dgram, float* dprob, void* stream);
/* igcn_head_loss_fwd and igcn_gram_loss_fwd_rbf_unit(out = NULL: the row partials stay in gscratch [Bg, 2 groups]) as two
 * roles of ONE launch: the two read disjoint outputs of the launches in front of them and write disjoint buffers. */
int igcn_head_loss_gram_fwd(int B, int K, int C, int NR, int S, const float* x1, const float* keep1, const float* W1,
                            const float* b1, const float* x2, const float* keep2, const float* W2, const float* b2,
                            const int64_t* y, const float* clin, const float* x_hat, const float* snps,
                            const float* lam6 /*HOST [6]*/, float hp_ce, float hp_mi, float* logp_out, float* reg_out,
                            float* dx1, float* dx2, float* dxhat, float* parts, float* wpart, float* dgram, float* dprob,
                            int Bg, int RD, int groups, const float* G, const float* tsne, int T, float gamma,
                            float* lap_out, float* gscratch, const float* gout /*HOST [2 groups]*/, float* Ssym,
                            void* stream);
int igcn_loss_final(const float* parts, int nparts, const float* gram, int gram_rows, const float* prob, int prob_rows,
                    const float* wts, float* out, void* stream);
/* igcn_loss_head_fwd that also writes the gradients igcn_loss_head_bwd would return for gout = 1 (each element's gradient
 * is known where its forward term is computed): a train step, whose d loss / d loss is one, then has no backward launch
 * for the loss head.  dlogp [2B,C] (of the raw scores when from_logits), dreg [2B,NR], dxhat [2B,S], dgram [4], dprob [1]. */
int igcn_loss_head_fwd_grads(int B, int C, int NR, int S, const float* logp, int from_logits, float* logp_out,
                             const int64_t* y, const float* reg, const float* clin, const float* x_hat, const float* snps,
                             const float* gram, int gram_rows, const float* prob, int prob_rows, const float* lam6,
                             float hp_ce, float hp_mi, float* loss, float* terms, float* dlogp, float* dreg, float* dxhat,
                             float* dgram, float* dprob, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Graph-diffusion pre-transform + block-diagonal collation of B dense adjacencies A [B,R,R] (f32), on the device —
 * replaces, per graph, preprocess_diffusion_imgs_snps (util_gdc.py:71-86: get_ppr_matrix :7-15 with `alpha`,
 * get_top_k_matrix :25-31 with `k`, scipy coo_matrix) and the index offsetting of Batch.from_data_list
 * (batch.py:98-104).  fp64 arithmetic in LDS, one workgroup per graph; R <= igcn_gdc_topk_max_rois().
 *   edge_index int64 [2, B*R*k], edge_attr f32 [B*R*k]: graph g owns slots [g*R*k, (g+1)*R*k), its edges in
 *   (row, col) order with both endpoints offset by g*R, then (-1,-1,0) padding if fewer than R*k entries are
 *   non-zero; counts int32 [B] = edges of each graph.
 */
int igcn_gdc_topk_max_rois(void);
int igcn_gdc_topk(int B, int R, int k, double alpha, const float* A, int64_t* edge_index, float* edge_attr,
                  int32_t* counts, void* stream);
/* The same for a batch DRAWN from a resident dataset A [S,R,R]: graph g of the batch is matrix subject[g] (int64 [B],
 * device) — a per-step producer needs no gathered copy of the B matrices.  A subject[g] outside [0, n_subjects) is
 * not read: graph g comes out empty (counts[g] = 0, all its slots padding). */
int igcn_gdc_topk_of(int B, int R, int k, double alpha, const float* A, int64_t n_subjects, const int64_t* subject,
                     int64_t* edge_index, float* edge_attr, int32_t* counts, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* IGCN_H */
