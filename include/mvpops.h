/*
 * mvpops.h -- C ABI of libmvpops.so, the MI355X (gfx950) point-cloud op layer.
 *
 * This is the native drop-in boundary for the MVP_Benchmark hot path.  The
 * reference reaches its CUDA kernels through 8 pybind11 modules whose bodies
 * immediately unwrap at::Tensor into raw pointers + int sizes + a stream and
 * call a `*_kernel_launcher` (e.g. utils/mm3d_pn2/ops/ball_query/src/
 * ball_query.cpp:30-43).  Each entry point below replaces one of those
 * launchers / pybind functions; the reference interface it replaces is cited
 * as path:line relative to the reference tree.
 *
 * Conventions (all entry points):
 *   - every pointer is a DEVICE pointer (hipMalloc / torch CUDA tensor
 *     data_ptr) to a contiguous row-major array; float = IEEE binary32,
 *     int = int32.  No torch types cross this boundary.
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream).
 *     Launches are asynchronous on that stream; nothing here synchronises.
 *   - the caller owns every buffer, including scratch; the library allocates
 *     nothing and keeps no DATA between calls (re-entrant, any thread).  The one
 *     piece of process-wide state is the set of tuning knobs behind
 *     mvp_emd_configure (which kernels mvp_emd_forward launches, never what it
 *     computes); nothing else is remembered.
 *   - return value: MVP_OK (0) on success; MVP_EBADSHAPE / MVP_EBADARG for
 *     argument errors (nothing launched); MVP_ELAUNCH if HIP reported a
 *     launch error (details via mvp_last_hip_error()).  The reference's
 *     `printf + return code that Python ignores` / `exit(-1)` behaviour
 *     (chamfer3D.cu:145-152, furthest_point_sample_cuda.cu:204-208) becomes
 *     a return code the host side turns into an exception.
 *   - initial contents required of output/scratch buffers are stated per
 *     function; they are exactly what the reference's Python wrappers
 *     establish before calling into native code.
 */
#ifndef MVPOPS_H_
#define MVPOPS_H_

#ifdef __cplusplus
extern "C" {
#endif

#define MVP_OK 0
#define MVP_EBADSHAPE (-1)
#define MVP_EBADARG (-2)
#define MVP_ELAUNCH (-3)

/* ABI version of this header; bumped on any signature change. */
#define MVP_ABI_VERSION 18
int mvp_abi_version(void);

/* hipGetErrorString of the last launch failure seen on the calling thread
 * ("" if none). */
const char *mvp_last_hip_error(void);

/* ---------------------------------------------------------------- Chamfer */

/* Replaces chamfer_3D.forward = chamfer_forward
 * (utils/metrics/CD/chamfer3D/chamfer_cuda.cpp:17-19,31) ->
 * chamfer_cuda_forward (chamfer3D.cu:136-154) -> NmDistanceKernel x2
 * (chamfer3D.cu:12-134).
 * xyz1 (b,n,3), xyz2 (b,m,3) -> dist1 (b,n), idx1 (b,n): squared distance to
 * and index of the nearest point of xyz2; dist2/idx2 (b,m): roles swapped.
 * Ties: lowest index wins.  Outputs are fully overwritten (n,m >= 1). */
int mvp_chamfer_forward(int b, int n, int m, const float *xyz1,
                        const float *xyz2, float *dist1, float *dist2,
                        int *idx1, int *idx2, void *stream);

/* Same contract and bit-identical results as mvp_chamfer_forward, for large
 * clouds: both sides are first bucketed into Morton-ordered cells inside the
 * caller's scratch (mvp_chamfer_scratch_bytes(b,n,m) bytes, 16-byte aligned,
 * contents irrelevant), then candidate tiles whose bounding box is farther
 * than the current best of a whole wave of queries are skipped.  Falls back
 * to the exhaustive kernel (scratch unused, may be NULL) when n or m < 2048
 * or n*m < 2^24. */
long long mvp_chamfer_scratch_bytes(int b, int n, int m);
int mvp_chamfer_forward_sorted(int b, int n, int m, const float *xyz1,
                               const float *xyz2, float *dist1, float *dist2,
                               int *idx1, int *idx2, void *scratch,
                               long long scratch_bytes, void *stream);

/* Replaces chamfer_3D.backward = chamfer_backward (chamfer_cuda.cpp:22-27,32)
 * -> chamfer_cuda_backward (chamfer3D.cu:176-195) -> NmDistanceGradKernel x2
 * (chamfer3D.cu:155-174).
 * gradxyz1 (b,n,3) and gradxyz2 (b,m,3) are ACCUMULATED into (float atomics)
 * and must be zero on entry (dist_chamfer_3D.py:56-57). */
int mvp_chamfer_backward(int b, int n, int m, const float *xyz1,
                         const float *xyz2, float *gradxyz1, float *gradxyz2,
                         const float *graddist1, const float *graddist2,
                         const int *idx1, const int *idx2, void *stream);

/* -------------------------------------------------------------------- EMD */

/* Bytes of device scratch mvp_emd_forward needs for (b, n); the reference
 * passes 11 scratch tensors instead (utils/metrics/EMD/emd_module.py:54-65).
 * Contents on entry are irrelevant (the kernel initialises its own state).
 * The last 16*b bytes of the buffer passed to mvp_emd_forward (its
 * scratch_bytes, a multiple of 16) receive per-cloud {int64 rounds, int64 bids}
 * statistics. */
long long mvp_emd_scratch_bytes(int b, int n);

/* Replaces emd.forward = emd_forward (utils/metrics/EMD/emd.cpp:14-20,29) ->
 * emd_cuda_forward (emd_cuda.cu:228-282): `iters` auction rounds of
 * {clear, calc_unass_cnt, calc_unass_cnt_sum, calc_unass_idx, Bid, GetMax,
 * Assign} (emd_cuda.cu:23-215) then CalcDist (:217-226), with the initial
 * state of emd_module.py:54-65.
 * xyz1 = prediction (b,n,3), xyz2 = ground truth (b,n,3) ->
 * dist (b,n) squared matched distance, assignment (b,n) index into xyz2
 * (may be non-injective after the forced last round).
 * Guards as emd_cuda.cu:236-249: n %% 1024 == 0, b <= 512 (-> MVP_EBADSHAPE);
 * iters >= 1; eps > 0 (-> MVP_EBADARG: the auction needs strictly positive bid
 * increments, and the search prunes on prices that never fall).  Deterministic: GetMax's racy last-writer (emd_cuda.cu:188-191)
 * is pinned to the highest qualifying bidder index.
 * Launches: one small memset (barrier words, hand-over records, statistics),
 * then a persistent cooperative kernel in which up to 8 workgroups share a
 * cloud when b leaves CUs free (b*W <= CU count), and -- for auctions of more
 * than 64 rounds, unless mvp_emd_configure(split = 0) -- a second cooperative
 * kernel of the same shape (csrc/emd_lean.hip) that takes a cloud over for the
 * rounds in which every workgroup has fewer bidders than four per wave (from
 * round ~100 on at 16384 points); it exits at once for clouds the first kernel
 * finished.  With split >= 2, 33..64 clouds of at least 4096 points on
 * four workgroups each, that second kernel stops before round 300 and a third launch runs the
 * rest with the workgroups dealt out again: the clouds with the most persons
 * still unassigned -- the ones whose rounds cost most -- get 8, the lightest 2
 * (64 uniform clouds: 8,5,4,4,3,3,3,2 over the eight clouds of an XCD; equally
 * loaded clouds keep 4 each)
 * (csrc/emd_lean.hip, emd_lean_tiers_kernel).  With split >= 4 (3: in a launch of its own)
 * clouds of at most 4096 points leave the clustered kernels as soon as at most
 * `resident_cap` (16) persons are unassigned -- round ~300 of 3000 at 1024
 * points, 500-850 at 2048, 900-1700 at 4096 -- and member 0 of the cloud's cluster
 * (csrc/emd_resident.h, one workgroup per cloud) runs the remaining rounds
 * with the whole auction state in that workgroup's LDS: no global memory access
 * inside a round.  With split = 5 (the default) the clustered kernels switch to
 * gathered-bid rounds once at most 256 persons of a cloud (of <= 16384 points)
 * are unassigned: a bid is published as a tagged 16-byte record, every workgroup
 * of the cluster settles every bid on its own view of the cloud (owner map in
 * LDS) and a round has one cluster-wide wait instead of a bid atomic and two
 * all-gathers (csrc/emd_lean.hip).  Once at most 16 persons of a cloud of more than
 * 4096 (and at most 16384) points are unassigned -- most of the 3000 rounds when the
 * prediction is near its ground truth -- member 0 of the cluster finishes the
 * auction alone: a wave per bidder with its record in registers, the round's bids and
 * the owner map in LDS, one barrier between Bid and Assign
 * (csrc/emd_lean_round_few.inc).  Which workgroups serve a cloud, in which
 * launch and in which kind of round, never changes a bit of the result.
 * If a cluster wait is abandoned (members not co-resident for tens of seconds;
 * never seen) dist is filled with NaN, assignment with -1 and the statistics
 * word `rounds` is negative: the host wrapper checks for NaN lazily, and
 * mvp_emd_backward skips negative indices. */
int mvp_emd_forward(int b, int n, const float *xyz1, const float *xyz2,
                    float *dist, int *assignment, float eps, int iters,
                    void *scratch, long long scratch_bytes, void *stream);

/* Tuning / A-B knobs of mvp_emd_forward, process-wide (compiled-in defaults; the release
 * library reads NOTHING from the environment -- the MVP_EMD_* variables of rounds 1-4 exist
 * only in libmvpops_hooks.so, the same sources built with -DMVP_TEST_HOOKS for the tests and
 * A/B tools).  A negative argument leaves that knob unchanged.
 *   cluster     0 = automatic, or 1|2|4|8: cap of the workgroups per cloud
 *   same_xcd    0: keep write-through stores even when a cluster shares an XCD
 *   split       5 (default): as 4, with gathered-bid rounds once <= 256 persons are unassigned;
 *               4: as 2, and clouds of <= 4096 points finish LDS-resident
 *               (csrc/emd_resident.h) on member 0 of their cluster, inside the second kernel's launch;
 *               3: the same in a launch of its own (every cloud waits for the last to get there); 2: the tail rounds run in the second kernel, from
 *               round 300 on with cluster widths by load (8 .. 2 workgroups);
 *               1: second kernel, fixed widths; 0: the first kernel runs every round
 *   resident_cap  1..64 (default 16: one wave of the workgroup per unassigned person; more: several per wave): unassigned persons at which a cloud of <= 4096
 *               points moves into LDS (split >= 3)
 * This is the library's only process-wide state (kept under a mutex; a call of
 * mvp_emd_forward reads one consistent copy).  Results never depend on it
 * (every setting is bit-identical: tests/test_gpu_ops.py).  A caller that must not
 * share state (several devices / threads with different plans) passes the plan on
 * the call instead: mvp_emd_forward_plan below. */
int mvp_emd_configure(int cluster, int same_xcd, int split, int resident_cap);

/* mvp_emd_forward with its launch plan on the call (ABI 17): the fields mean what
 * mvp_emd_configure's arguments mean; a NULL plan or a negative field selects the
 * compiled-in default.  Neither reads nor writes the process-wide knobs.  Same
 * results bit for bit. */
typedef struct MvpEmdPlan {
  int cluster, same_xcd, split, resident_cap;
} MvpEmdPlan;
int mvp_emd_forward_plan(int b, int n, const float *xyz1, const float *xyz2,
                         float *dist, int *assignment, float eps, int iters,
                         void *scratch, long long scratch_bytes,
                         const MvpEmdPlan *plan, void *stream);

/* Replaces emd.backward = emd_backward (emd.cpp:22-25,30) ->
 * emd_cuda_backward (emd_cuda.cu:302-316) -> NmDistanceGradKernel (:284-300).
 * gradxyz (b,n,3) accumulated into; must be zero on entry
 * (emd_module.py:77). */
int mvp_emd_backward(int b, int n, const float *xyz1, const float *xyz2,
                     float *gradxyz, const float *graddist, const int *idx,
                     void *stream);

/* -------------------------------------------------- furthest point sample */

/* Replaces furthest_point_sample_ext.furthest_point_sampling_wrapper
 * (utils/mm3d_pn2/ops/furthest_point_sample/src/furthest_point_sample.cpp:
 * 32-43,61) -> furthest_point_sampling_kernel_launcher
 * (src/furthest_point_sample_cuda.cu:143-209).
 * points (b,n,3); temp (b,n) scratch (the reference requires it pre-filled
 * with 1e10, furthest_point_sample.py:30; this implementation does not read
 * it on entry but leaves the final min-distances in it); idx (b,m) out.
 * Tie order is the reference's: strided first-strict-max per thread then the
 * shared-memory tree of furthest_point_sample_cuda.cu:17-23. */
int mvp_furthest_point_sampling(int b, int n, int m, const float *points,
                                float *temp, int *idx, void *stream);

/* Same contract and index-identical results as mvp_furthest_point_sampling for
 * 4096 < n <= 16384 (elsewhere it simply forwards to it): the cloud is first
 * Morton-sorted into the caller's scratch (mvp_fps_scratch_bytes(b,n) bytes,
 * 16-byte aligned, contents irrelevant) so that a lane owns a compact run of
 * points and whole waves skip the distance update when the new sample cannot
 * lower any of their running minima. */
long long mvp_fps_scratch_bytes(int b, int n);
int mvp_furthest_point_sampling_sorted(int b, int n, int m, const float *points,
                                       float *temp, int *idx, void *scratch,
                                       long long scratch_bytes, void *stream);

/* Replaces furthest_point_sampling_with_dist_wrapper
 * (furthest_point_sample.cpp:45-58,63) -> ..._with_dist_kernel_launcher
 * (furthest_point_sample_cuda.cu:333-400).  points_dist (b,n,n). */
int mvp_furthest_point_sampling_with_dist(int b, int n, int m,
                                          const float *points_dist,
                                          float *temp, int *idx, void *stream);

/* The same sampling (same indices, same temp) with w = 2 or 4 workgroups per cloud: a lane owns
 * 1/w of the points the reference's thread scans, the members exchange their local winners
 * through memory once per round (csrc/fps.hip: fps_cluster_kernel).  n a multiple of w * 1024,
 * n <= 16384; scratch of mvp_fps_cluster_scratch_bytes(b) bytes.  MVP_EBADSHAPE when the shape is
 * not covered or the cooperative launch does not fit the device (the caller then uses
 * mvp_furthest_point_sampling).  Opt-in: measured slower or on par, see profiles/NOTES_r1-r3_design_notebook.md section 10. */
long long mvp_fps_cluster_scratch_bytes(int b);
int mvp_furthest_point_sampling_cluster(int b, int n, int m, int w, const float *points,
                                        float *temp, int *idx, void *scratch,
                                        long long scratch_bytes, void *stream);


/* ---------------------------------------------------- ball_query/knn/3nn */

/* Replaces ball_query_ext.ball_query_wrapper
 * (utils/mm3d_pn2/ops/ball_query/src/ball_query.cpp:30-43,46) ->
 * ball_query_kernel_launcher (src/ball_query_cuda.cu:56-78).
 * new_xyz (b,m,3) centres, xyz (b,n,3) -> idx (b,m,nsample).  Rows with no
 * hit keep the zeros the caller put there (ball_query.py:35); this
 * implementation writes every slot itself, so idx need not be pre-zeroed. */
int mvp_ball_query(int b, int n, int m, float min_radius, float max_radius,
                   int nsample, const float *new_xyz, const float *xyz,
                   int *idx, void *stream);

/* Replaces knn_ext.knn_wrapper (utils/mm3d_pn2/ops/knn/src/knn.cpp:28-41,45)
 * -> knn_kernel_launcher (src/knn_cuda.cu:97-115).
 * xyz (b,n,3), new_xyz (b,m,3) -> idx (b,m,nsample), dist2 (b,m,nsample),
 * ascending; 1 <= nsample <= 100.  Slot order follows the reference's
 * max-heap + heap-sort (knn_cuda.cu:26-53). */
int mvp_knn(int b, int n, int m, int nsample, const float *xyz,
            const float *new_xyz, int *idx, float *dist2, void *stream);

/* The same result as mvp_knn for large clouds without the exhaustive scan: both point sets are
 * Morton-sorted into caller scratch (mvp_knn_scratch_bytes(b, n, m) bytes, 16-byte aligned, contents
 * irrelevant) and every query searches outwards through boxed tiles, keeping k + 1 candidates; queries
 * whose k + 1 smallest distances are not pairwise different (lattices, duplicates: there the reference's
 * result depends on its sequence of heap operations, knn_cuda.cu:26-53,80-90) are recomputed by the
 * exhaustive kernel.  n < 4096, m < 1024 or nsample > 32: forwards to mvp_knn. */
long long mvp_knn_scratch_bytes(int b, int n, int m);
int mvp_knn_sorted(int b, int n, int m, int nsample, const float *xyz,
                   const float *new_xyz, int *idx, float *dist2, void *scratch,
                   long long scratch_bytes, void *stream);

/* Feature-space neighbour search of the models (no native counterpart in the
 * reference: completion/model_utils.py:242-247 builds `-xx - inner - xx^T` on a
 * materialised (B,N,N) matrix and calls torch.topk).  dot (b,n,n) = x^T x from a
 * library GEMM, sq (b,n) = |x_i|^2 -> idx (b,n,k): per row i the k largest of
 * fl(fl(-sq[j] + 2 dot[i][j]) - sq[i]), sorted descending (self first); equal
 * to torch.topk's indices wherever the values are distinct.  k <= n; k <= 64 and
 * n <= 16384: one wave per row streaming the matrix (equal values keep the lower
 * column first); otherwise the tile kernel, k <= 47 (MVP_EBADSHAPE beyond). */
int mvp_topk_gram(int b, int n, int k, const float *dot, const float *sq,
                  int *idx, void *stream);

/* Replaces interpolate_ext.three_nn_wrapper
 * (utils/mm3d_pn2/ops/interpolate/src/interpolate.cpp:46-56,88) ->
 * three_nn_kernel_launcher (src/three_nn_cuda.cu:67-89).
 * unknown (b,n,3), known (b,m,3) -> dist2 (b,n,3) SQUARED, idx (b,n,3). */
int mvp_three_nn(int b, int n, int m, const float *unknown,
                 const float *known, float *dist2, int *idx, void *stream);

/* ----------------------------------------- interpolate / gather / group */

/* Replaces interpolate_ext.three_interpolate_wrapper (interpolate.cpp:58-70,
 * 89) -> three_interpolate_kernel_launcher
 * (src/three_interpolate_cuda.cu:37-59).
 * points (b,c,m), idx (b,n,3), weight (b,n,3) -> out (b,c,n). */
int mvp_three_interpolate(int b, int c, int m, int n, const float *points,
                          const int *idx, const float *weight, float *out,
                          void *stream);

/* Replaces interpolate_ext.three_interpolate_grad_wrapper
 * (interpolate.cpp:72-85,91) -> three_interpolate_grad_kernel_launcher
 * (three_interpolate_cuda.cu:86-108).  grad_points (b,c,m) accumulated into;
 * zero on entry (three_interpolate.py:53). */
int mvp_three_interpolate_grad(int b, int c, int n, int m,
                               const float *grad_out, const int *idx,
                               const float *weight, float *grad_points,
                               void *stream);

/* Replaces gather_points_ext.gather_points_wrapper
 * (utils/mm3d_pn2/ops/gather_points/src/gather_points.cpp:28-38,55) ->
 * gather_points_kernel_launcher (src/gather_points_cuda.cu:28-49).
 * points (b,c,n), idx (b,npoints) -> out (b,c,npoints). */
int mvp_gather_points(int b, int c, int n, int npoints, const float *points,
                      const int *idx, float *out, void *stream);

/* Gather + max over the k neighbours of every output point, fused (not an operator of the
 * reference: it replaces the gather_points + torch.max pair of edge_preserve_sampling,
 * completion/model_utils.py:101-104, without materialising the (b,c,npoints,k) neighbour tensor):
 *   out[b,c,p] = max_j points[b,c, idx[b,p,j]],  arg[b,c,p] = idx[b,p,j*] for the FIRST maximal j.
 * points (b,c,n), idx (b,npoints,k) int32 in [0,n), out (b,c,npoints), arg (b,c,npoints) int32.
 * MVP_EBADSHAPE when a row of n floats does not fit the 96 KiB LDS staging buffer (n > 24576): the
 * caller then gathers and reduces. */
int mvp_gather_max(int b, int c, int n, int npoints, int k, const float *points,
                   const int *idx, float *out, int *arg, void *stream);

/* Its gradient: grad_points[b,c, arg[b,c,p]] += grad_out[b,c,p] (overwrite != 0: every element of
 * grad_points is written, no zero fill by the caller). */
int mvp_gather_max_grad(int b, int c, int n, int npoints, const float *grad_out,
                        const int *arg, float *grad_points, int overwrite, void *stream);

/* Replaces gather_points_ext.gather_points_grad_wrapper
 * (gather_points.cpp:40-52,57) -> gather_points_grad_kernel_launcher
 * (gather_points_cuda.cu:72-94).  grad_points (b,c,n) accumulated into; zero
 * on entry (gather_points.py:44). */
int mvp_gather_points_grad(int b, int c, int n, int npoints,
                           const float *grad_out, const int *idx,
                           float *grad_points, void *stream);

/* Replaces group_points_ext.forward
 * (utils/mm3d_pn2/ops/group_points/src/group_points.cpp:45-57,60) ->
 * group_points_kernel_launcher (src/group_points_cuda.cu:81-101).
 * points (b,c,n), idx (b,npoints,nsample) -> out (b,c,npoints,nsample). */
int mvp_group_points(int b, int c, int n, int npoints, int nsample,
                     const float *points, const int *idx, float *out,
                     void *stream);

/* Replaces group_points_ext.backward (group_points.cpp:31-43,61) ->
 * group_points_grad_kernel_launcher (group_points_cuda.cu:33-54).
 * grad_points (b,c,n) accumulated into; zero on entry
 * (group_points.py:213). */
int mvp_group_points_grad(int b, int c, int n, int npoints, int nsample,
                          const float *grad_out, const int *idx,
                          float *grad_points, void *stream);

/* ---- the three scatter-add gradients with caller-provided scratch.
 * Same contracts as mvp_gather_points_grad / mvp_group_points_grad /
 * mvp_three_interpolate_grad (same reference launchers; grad_points is
 * accumulated into).  The index array is inverted once per call into `scratch`
 * (counting sort by destination per cloud and chunk of grad_out columns), then
 * every destination is summed by one thread from LDS-staged grad_out columns:
 * no float atomics, grad_out read once.  mvp_scatter_scratch_bytes(b, n_dst,
 * m_src, r): bytes needed for b clouds, n_dst destination points (the n of
 * gather/group, the m of three_interpolate), m_src grad_out columns (npoints,
 * npoints*nsample, or three_interpolate's n) and r index entries per column
 * (1, or 3 for three_interpolate); 0 = shape not covered (n_dst > 8192).  With
 * scratch == NULL, scratch_bytes too small or a shape that is not covered the
 * _ws entry points run the plain ones.
 * mode: bit 0 MVP_SCATTER_OVERWRITE -- grad_points is WRITTEN (every element;
 * it need not be zero on entry and is not read), saving the caller's zero fill
 * and the read-modify-write; bit 1 MVP_SCATTER_INDEX_READY -- `scratch` still
 * holds the inverted index a previous call built for exactly this idx (and
 * weight): the sort is skipped (the same neighbour graph is differentiated
 * through several gathers per step). */
#define MVP_SCATTER_OVERWRITE 1
#define MVP_SCATTER_INDEX_READY 2
long long mvp_scatter_scratch_bytes(int b, int n_dst, int m_src, int r);
int mvp_gather_points_grad_ws(int b, int c, int n, int npoints,
                              const float *grad_out, const int *idx,
                              float *grad_points, void *scratch,
                              long long scratch_bytes, int mode, void *stream);
int mvp_group_points_grad_ws(int b, int c, int n, int npoints, int nsample,
                             const float *grad_out, const int *idx,
                             float *grad_points, void *scratch,
                             long long scratch_bytes, int mode, void *stream);
int mvp_three_interpolate_grad_ws(int b, int c, int n, int m,
                                  const float *grad_out, const int *idx,
                                  const float *weight, float *grad_points,
                                  void *scratch, long long scratch_bytes,
                                  int mode, void *stream);

/* Shared-weight neighbourhood aggregation of VRCNet's point self-attention (no
 * native counterpart in the reference: completion/models/vrcnet.py:52-55
 * materialises w.repeat(1, share, 1, 1), the product and its sum over k).
 * w (b,cw,k,n), v (b,share*cw,k,n) -> out (b,share*cw,n):
 *   out[b, s*cw+m, n] = sum_k w[b,m,k,n] * v[b, s*cw+m, k, n]   (k ascending).
 * share in {1,2,4,8,16} (MVP_EBADSHAPE otherwise). */
int mvp_share_weighted_sum(int b, int share, int cw, int k, int n,
                           const float *w, const float *v, float *out,
                           void *stream);
/* Its gradients in one pass: grad_v (b,share*cw,k,n) = w * grad_out (broadcast
 * over k), grad_w (b,cw,k,n) = sum_s grad_out[b,s*cw+m,n] * v[b,s*cw+m,k,n];
 * both overwritten. */
int mvp_share_weighted_sum_grad(int b, int share, int cw, int k, int n,
                                const float *w, const float *v,
                                const float *grad_out, float *grad_w,
                                float *grad_v, void *stream);

/* Weight (and bias) gradient of a per-point / per-edge linear map -- a 1x1
 * convolution y = W x + bias on x (b,cin,len), len = N or k*N positions per
 * cloud -- for few output channels (the models' Conv1d/Conv2d(kernel_size=1)
 * layers of completion/models; the reference leaves them to cuDNN):
 *   gw[co][ci] = sum_{b,l} gy[b][co][l] * x[b][ci][l]   (overwritten)
 *   gb[co]     = sum_{b,l} gy[b][co][l]                 (overwritten; gb may be NULL)
 * cout <= 64, len % 4 == 0, x and gy 16-byte aligned.  scratch:
 * mvp_pointwise_wgrad_scratch_bytes(b, cin, cout, len) bytes of per-workgroup
 * partial sums (0 = shape not covered), added up in a fixed order. */
long long mvp_pointwise_wgrad_scratch_bytes(int b, int cin, int cout, int len);
int mvp_pointwise_wgrad(int b, int cin, int cout, int len, const float *x,
                        const float *gy, float *gw, float *gb, void *scratch,
                        long long scratch_bytes, void *stream);

/* Data gradient of the same layers:  gx[b][ci][l] = sum_co weight[co][ci] * gy[b][co][l]
 * (overwritten); cin, cout <= 64, len % 4 == 0, gy and gx 16-byte aligned.  The callers'
 * ReLU mask is applied to gy beforehand (mvp_benchmark_amd/pointwise.py). */
int mvp_pointwise_dgrad(int b, int cin, int cout, int len, const float *weight,
                        const float *gy, float *gx, void *stream);

/* The same aggregation with the gather of the neighbours' values fused in:
 *   out[b, s*cw + m, p] = sum_k w[b, m, k, p] * v[b, s*cw + m, idx[b, k, p]]
 * w (b,cw,k,n), v (b,share*cw,n_src) the per-point values, idx (b,k,n) int32 the neighbour lists (k-major);
 * the (b, share*cw, k, n) tensor of gathered values (vrcnet.py:45-55 builds it with get_edge_features) is never
 * formed.  share * n_src * 4 <= 96 KiB (mvp_share_gather_sum_lds_bytes); bit-identical to mvp_group_points +
 * mvp_share_weighted_sum.  _grad: grad_w (b,cw,k,n) and grad_vals (b,share*cw,k,n) = the gradient of the gathered
 * values, to be scattered into grad_v by mvp_group_points_grad(_ws). */
long long mvp_share_gather_sum_lds_bytes(int share, int n_src);
int mvp_share_gather_sum(int b, int share, int cw, int k, int n_src, int n, const float *w,
                         const float *v, const int *idx, float *out, void *stream);
int mvp_share_gather_sum_grad(int b, int share, int cw, int k, int n_src, int n, const float *w,
                              const float *v, const int *idx, const float *grad_out,
                              float *grad_w, float *grad_vals, void *stream);

/* ------------------------------------------- grouped-feature MLP on MFMA */

/* 1x1 convolution / per-point linear map with a fused epilogue, on the matrix
 * cores (v_mfma_f32_32x32x2_f32: float32 in, float32 accumulate -- the result is
 * a k-ordered fmaf chain, no reduced precision).  No native counterpart in the
 * reference, which calls cuDNN through nn.Conv1d / nn.Conv2d(kernel_size=1) and
 * separate bias / ReLU / add / max kernels (completion/model_utils.py:26-55,
 * completion/models/vrcnet.py:21-57, ecg.py:36-65, pcn.py).
 *   x (b, cin, len), w (cout, cin) [w_kmajor = 0] or (cin, cout) [w_kmajor = 1],
 *         its rows ldw floats apart (0 = dense; a (cout, cin) weight with cin % 4 != 0
 *         is read with 16-byte loads when the caller pads its rows to ldw % 4 == 0)
 *   xmask (b, cin, len) or NULL: x is taken as 0 where xmask <= 0 (the data
 *         gradient of a fused ReLU: x = grad_out, xmask = the layer's output)
 *   t[b,co,l] = sum_ci w(co,ci) x[b,ci,l] + bias[co]        (bias may be NULL)
 *   t = max(t, 0)                                           if relu
 *   u[b,co,g] = max_{l in group g} t[b,co,l]                groups of `group`
 *                                                            consecutive columns
 *                                                            (1, 2, 4, ..., 32)
 *   y[b,co,g] = u + residual[b,co,g]                        (residual may be NULL)
 * y, residual: (b, cout, len / group).  len % 4 == 0, len % group == 0, x 16-byte
 * aligned.  The data gradient of the plain map is the same call with the forward
 * weight, cin and cout swapped and w_kmajor = 1. */
int mvp_pointwise_mfma(int b, int cin, int cout, int len, const float *x,
                       const float *xmask, const float *w, int ldw, int w_kmajor,
                       const float *bias, const float *residual, int relu,
                       int group, float *y, void *stream);

/* ABI 18.  The same map with the activation / residual patterns of the relational encoder in the GEMM's
 * prologue and epilogue, so that `conv(relu(x))`, `relu(conv(a) + x)`, `relu(conv(f) + g[b])` and two
 * convolutions of one input cost no elementwise pass and no copy (completion/models/vrcnet.py:34-36,54-57
 * pre-activation ReLUs of SA_module; :151,172 and :255-296 residual sums followed by ReLU; :283-285 conv6 over
 * cat(global feature tiled, features); :160,172 conv1 / conv_res of one input).  mvp_pointwise_mfma is this call
 * with bias_per_cloud = 0, flags = relu ? MVP_PW_RELU : 0, m_split = 0.
 *   flags  MVP_PW_X_RELU       x counts as max(x, 0) (on load; x itself is not written)
 *          MVP_PW_RELU         t = max(t, 0) before the group maximum and the residual (mvp_pointwise_mfma's relu)
 *          MVP_PW_RES_IS_MASK  `residual` is a mask: y = residual > 0 ? u : 0 instead of u + residual -- the data
 *                              gradient of conv(relu(x)): W^T grad_out where x > 0
 *          MVP_PW_RELU_AFTER   y = max(y, 0) after the residual / mask
 *   bias_per_cloud != 0: bias is (b, cout) -- the share of a per-cloud vector in a convolution over
 *          cat(vector tiled over the positions, features) -- instead of (cout)
 *   m_split (0, or a multiple of 32 below cout; then residual == NULL, group == 1, y2 != NULL): output rows below
 *          m_split go to y (b, m_split, len), the others to y2 (b, cout - m_split, len).
 * Arithmetic: the same k-ordered fmaf chain per output as mvp_pointwise_mfma; every fused step is the exact
 * float operation the separate pass would perform (max, add, select). */
#define MVP_PW_RELU 1
#define MVP_PW_RELU_AFTER 2
#define MVP_PW_RES_IS_MASK 4
#define MVP_PW_X_RELU 8
int mvp_pointwise_mfma_ex(int b, int cin, int cout, int len, const float *x,
                          const float *xmask, const float *w, int ldw, int w_kmajor,
                          const float *bias, int bias_per_cloud, const float *residual,
                          int flags, int group, float *y, int m_split, float *y2,
                          void *stream);

/* ABI 16.  (W x + bias) [then ReLU], reduced with max over ALL positions of a cloud inside the GEMM's epilogue
 * -- the PointNet stage of the completion networks: `x = self.conv4(x); global_feature, _ = torch.max(x, 2)`
 * (completion/models/pcn.py:29-30; vrcnet.py:281-282 conv5; ecg.py:137-138 gf_conv) -- without writing the
 * (b, cout, len) tensor: val (b, cout) = the maxima, idx (b, cout) int32 = the first position that attains each
 * (what torch.max reports; the sparse backward pass mvp_pointwise_max_backward takes it).  Arithmetic as
 * mvp_pointwise_mfma: val equals its output reduced with max, bit for bit.  keys: b * cout * 8 bytes of caller
 * scratch, 8-byte aligned, contents irrelevant.  w (cout, cin) row-major with row stride ldw (0 = cin). */
int mvp_pointwise_mfma_max(int b, int cin, int cout, int len, const float *x, const float *w, int ldw,
                           const float *bias, int relu, float *val, int *idx, void *keys, long long keys_bytes,
                           void *stream);

/* Weight (and bias) gradient of the same map on the matrix cores:
 *   gw[co,ci] = sum_b sum_l g[b,co,l] x[b,ci,l],  gb[co] = sum_b sum_l g[b,co,l]
 * with g = gy, or gy where gymask > 0 and 0 elsewhere (fused ReLU).  gb may be
 * NULL.  The b*len positions are split over workgroups that write partial
 * tiles into `scratch` (mvp_pointwise_wgrad_mfma_scratch_bytes(...) bytes; 0 =
 * shape not covered), a second kernel adds them in a fixed order: no float
 * atomics, bit-reproducible.  gw, gb are overwritten.  len % 4 == 0, operands
 * 16-byte aligned. */
long long mvp_pointwise_wgrad_mfma_scratch_bytes(int b, int cin, int cout,
                                                 int len, int with_bias);
int mvp_pointwise_wgrad_mfma(int b, int cin, int cout, int len, const float *x,
                             const float *gy, const float *gymask, float *gw,
                             float *gb, void *scratch, long long scratch_bytes,
                             void *stream);
/* ABI 18.  x_relu != 0: x counts as max(x, 0) on load -- the weight gradient of conv(relu(x)) from the
 * tensor the layer was handed, not from a stored relu(x). */
int mvp_pointwise_wgrad_mfma_ex(int b, int cin, int cout, int len, const float *x, int x_relu,
                                const float *gy, const float *gymask, float *gw,
                                float *gb, void *scratch, long long scratch_bytes,
                                void *stream);

/* Backward pass of a 1x1 convolution followed by a max over the positions,
 * v[b][co] = max_l (W x + bias)[b][co][l] (the PointNet stage: completion/models/pcn.py:25-31,
 * conv5 of vrcnet.py's relational encoder, ecg.py's gf_conv), through the b * cout winning
 * positions only:
 *   gw[co][ci]   = sum_b g[b][co] x[b][ci][idx[b][co]],   gb[co] = sum_b g[b][co]
 *   gx[b][ci][l] = sum_{co : idx[b][co] = l} g[b][co] w[co][ci]     (0 elsewhere; every entry written)
 * x (b,cin,len), w (cout,cin), g (b,cout) = the gradient of v, idx (b,cout) int32 = the position
 * torch.max reported.  gx, gw may each be NULL (that gradient is not needed), gb too.
 * len <= 16384, cout <= 4096.  Fixed summation orders: bit-reproducible.  scratch (may be NULL):
 * mvp_pointwise_max_backward_scratch_bytes(...) bytes (0 = shape not covered by the staged pass) in which the
 * weight gradient stages the b * cout winning columns of x with coalesced accesses; without it the columns are
 * gathered directly (a 64-byte sector per value).  The reference lets cuDNN run its dense data- and
 * weight-gradient passes on the max's gradient, a tensor of zeros. */
long long mvp_pointwise_max_backward_scratch_bytes(int b, int cin, int cout, int len);
int mvp_pointwise_max_backward(int b, int cin, int cout, int len, const float *x,
                               const float *w, const float *g, const int *idx,
                               float *gx, float *gw, float *gb, void *scratch,
                               long long scratch_bytes, void *stream);

/* ------------------------------------------------ registration (DCP) head */

/* Replaces the per-sample loop of SVDHead.forward
 * (registration/models/dcp.py:360-373, registration/model_utils.py:229-240):
 *   u, s, v = torch.svd(H[i]); r = v @ u.T; if det(r) < 0: r = (v @ diag(1,1,-1)) @ u.T
 * for all b matrices in one launch, no host synchronisation.
 * H (b,3,3) row-major -> R (b,3,3) the rotation (reflection fix applied);
 * optional (may be NULL): U (b,3,3), S (b,3) descending, V (b,3,3) with
 * H = U diag(S) V^T (V WITHOUT the fix), flipped (b) = 1 where the fix applied.
 * One-sided Jacobi in float64 per matrix, outputs rounded to float32. */
int mvp_kabsch_svd3(int b, const float *H, float *R, float *U, float *S,
                    float *V, int *flipped, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* MVPOPS_H_ */
