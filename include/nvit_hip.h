/* nvit_hip.h — C ABI of the MI355X (gfx950) nViT hot-path library (libnvit_hip.so).
 *
 * Drop-in boundary: the reference has no FFI; its seams are the Python
 * classes in /root/reference/nvit/model.py and Trainer.normalize_matrices
 * (/root/reference/nvit/train.py:461-480).  Every entry point below replaces the torch
 * operator sequence cited next to it.  Conventions:
 *   - plain pointers and sizes only; all pointers are DEVICE pointers unless marked host;
 *   - the caller owns every buffer (outputs, workspaces, saved-for-backward tensors);
 *     the library never allocates, frees or keeps a pointer past the call;
 *   - all work is enqueued on `stream` (a hipStream_t passed as void*); calls are
 *     asynchronous and re-entrant (forward on the main thread, backward on the autograd
 *     thread); the only global state is the optional event-timing table (nvit_prof_*);
 *   - return 0 on success, NVIT_EINVAL for a rejected argument, else a hipError_t value;
 *     nvit_last_error() gives the calling thread's message.  No exception crosses the ABI.
 *   - `dt` selects the activation / MFMA-operand type: NVIT_F32 (exact-f32 MFMA path,
 *     parity <=1e-5) or NVIT_BF16 (bf16 operands, fp32 accumulate).  The residual stream,
 *     parameters, gradients and all reductions are fp32 in both modes.
 *   - token-major activations are [M, C] with M = B*T rows; per-head tensors are
 *     [B, H, T, d] contiguous.
 */
#ifndef NVIT_HIP_H
#define NVIT_HIP_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NVIT_OK 0
#define NVIT_EINVAL 1001
#define NVIT_F32 0
#define NVIT_BF16 1
#define NVIT_BF16_F32IN 3 /* nvit_qknorm_fwd / nvit_swiglu_fwd only: fp32 inputs (unrounded GEMM outputs), bf16 outputs */

int nvit_version(void);
const char* nvit_last_error(void);

/* ---- event timing of kernel families (bench.py roofline leg) ------------------ */
#define NVIT_KID_GEMM_NT 0
#define NVIT_KID_GEMM_TN 1
#define NVIT_KID_ATTN_FWD 2
#define NVIT_KID_ATTN_BWD 3
#define NVIT_KID_ROWOPS 4
#define NVIT_KID_RENORM 5
#define NVIT_KID_SHADOW 6
#define NVIT_KID_PATCHIFY 7
#define NVIT_KID_MISC 8
#define NVIT_KID_GEMM_F32 9 /* exact-f32 MFMA GEMMs (fp32 mode; patch embedding + classifier of the bf16 mode) */
#define NVIT_KID_GEMM_SWIGLU 10     /* bf16 NT GEMM with the SwiGLU epilogue (nvit_gemm_nt_swiglu) */
#define NVIT_KID_GEMM_QKNORM 11     /* bf16 NT GEMM with the q/k-normalise + head-split epilogue (nvit_gemm_nt_qknorm) */
#define NVIT_KID_GEMM_SWIGLU_BWD 12 /* bf16 NT GEMM with the SwiGLU-backward epilogue (nvit_gemm_nt_swiglu_bwd) */
#define NVIT_KID_OPTIM 13           /* nvit_grad_sqnorm + nvit_adamw_renorm (NVIT_KID_RENORM = stand-alone nvit_renorm_weights) */
#define NVIT_KID_COUNT 14
void nvit_prof_enable(int on);
/* time only the families whose bit is set (bit k = family k of nvit_prof_name); nvit_prof_enable(1) = all, (0) = none */
void nvit_prof_select(unsigned mask);
/* Synchronises the recorded events and returns, per kernel family, total milliseconds,
 * algorithmic FLOPs, algorithmic bytes and launch count since the last collect. Host arrays
 * of NVIT_KID_COUNT entries. */
int nvit_prof_collect(double* ms, double* flops, double* bytes, int64_t* launches);
const char* nvit_prof_name(int kid);

/* ---- weights --------------------------------------------------------------------
 * nvit_renorm_weights: Trainer.normalize_matrices (train.py:461-480) as ONE persistent
 * launch.  `table` is a device array of n rows of 5 int64: {ptr, rows, cols, dim, first_item}
 * dim=1: every row scaled to unit L2 norm (query/key/value/c_fc); dim=0: every column
 * (att_c_proj/mlp_c_proj).  fp32 in place, one read + one write per element.
 * first_item = prefix sum of work items (rows/NVIT_RENORM_ROWS_PER_ITEM for dim=1,
 * cols/NVIT_RENORM_COLS_PER_ITEM for dim=0, rounded up); total_items = sum.
 * dim=0 keeps a [rows x NVIT_RENORM_COLS_PER_ITEM] panel in the registers of a 1024-thread workgroup (rows <= 1152). */
#define NVIT_RENORM_ROWS_PER_ITEM 64
#define NVIT_RENORM_COLS_PER_ITEM 64
int nvit_renorm_weights(const int64_t* table, int n, int total_items, void* stream);

/* nvit_shadow_weights: builds the private MFMA-operand copies of the fp32 master weights
 * (SURVEY.md §8b "state_dict contract": non-persistent shadows) in ONE launch.  `table` =
 * device array of n rows of 12 int64:
 *   {src, rows, cols, dst, dst_ld, dst_cols, dstT, dstT_ld, dstT_cols, perm, first_item, tiles_c}
 *   dst  [rows, dst_ld]  : dst[r, c] = src[perm(r), c] for c < cols, 0 for cols <= c < dst_cols
 *   dstT [cols, dstT_ld] : dstT[c, r] = dst[r, c] for r < rows, 0 for rows <= r < dstT_cols
 *   Either dst or dstT may be 0 (skipped).  dst/dstT may point inside a larger concatenated
 *   buffer (Q|K|V stacking); only the stated extents are written.
 *   perm: 0 identity; 1 = SwiGLU interleave of a [2F, K] matrix: shadow row 32q+w is
 *         source row 16q+w (w<16, "u") or F+16q+(w-16) (w>=16, "v");
 *         2 = split-precision image (dt = bf16 only, dstT unused): dst [rows, 2*Kp], Kp = nvit_patch_embed_kp(cols);
 *         columns 32j..32j+31 of src become the 128-byte slice [hi32 | lo32] at dst column 64j, hi = bf16(src),
 *         lo = bf16(src - hi) - the weight operand of nvit_patch_embed_fwd.  The padding columns of the last
 *         slice are NOT written: allocate dst zeroed.
 *   Work items are 64x64 tiles: tiles_c = ceil(max(cols,dst_cols)/64) tile columns,
 *   ceil(max(rows,dstT_cols)/64) tile rows; first_item = prefix sum; total_items = sum.
 * Element type of dst/dstT is `dt`. */
int nvit_shadow_weights(const int64_t* table, int n, int total_items, int dt, void* stream);

/* Optimizer step fused with the re-normalisation: replaces clip_grad_norm_ + AdamW.step +
 * Trainer.normalize_matrices (train.py:935-946, 461-480, 989-990; parameter groups of model.py:369-385).
 * table: n rows of 10 int64 {p, g, m, v (fp32 device pointers), rows, cols, kind, first_item, first_chunk,
 *        f32bits(lr) | f32bits(weight_decay) << 32};  kind -1 = plain (items of 8192 elements), 1 = normalise rows
 *        (items of 16 rows; cols % 4 == 0, cols <= 1536), 0 = normalise columns (items of 32 columns, rows <= 1152);
 *        first_chunk counts 8192-element chunks of the gradient for nvit_grad_sqnorm.
 * nvit_grad_sqnorm: partial[npart] = per-workgroup sums of g*g over all gradients (npart <= 4096 workgroups).
 * nvit_adamw_renorm: clip = min(1, max_norm / (sqrt(sum partial) + 1e-6)) when partial != NULL and max_norm > 0;
 *        g *= clip; torch AdamW update (decoupled decay, bias corrections 1 - beta^t passed by the host); rows /
 *        columns of kind 1 / 0 matrices are L2-normalised before the single write-back.  gnorm_out[0] (optional)
 *        receives the pre-clip global gradient norm.  max_slab_rows = largest `rows` among kind-0 entries.
 *        hyper (optional, 3 floats on the device): when given, the bias corrections are read from hyper[1..2] as
 *        maintained by nvit_adamw_tick (hyper[0] = step count; one call per optimizer step, before this one), so the
 *        whole step can be captured in a hipGraph and replayed; the two double arguments are then ignored. */
int nvit_adamw_tick(float* hyper, double beta1, double beta2, void* stream);
int nvit_grad_sqnorm(const int64_t* table, int n, int total_chunks, float* partial, int npart, void* stream);
int nvit_adamw_renorm(const int64_t* table, int n, int total_items, int max_slab_rows, float beta1, float beta2,
                      float eps, double bias_correction1, double bias_correction2, const float* partial, int npart,
                      float max_norm, float* gnorm_out, const float* hyper, void* stream);

/* ---- GEMMs ------------------------------------------------------------------------
 * nvit_gemm_nt: C[M,N] = A[M,K] * B[N,K]^T  (nn.Linear: model.py:99-101,130,148,155,...)
 * A, B of type dt, K % (128/sizeof(dt)) == 0, lda/ldb multiples of 16 bytes.
 * Epilogue, in this order: +bias[n]; *colscale[n]; +rowadd[(m % rowadd_period)*N + n];
 * + old C (accumulate!=0); store as out_dt (NVIT_F32 / NVIT_BF16) with leading dim ldc. */
int nvit_gemm_nt(int dt, const void* A, int lda, const void* B, int ldb, void* C, int ldc, int out_dt,
                 int M, int N, int K, const float* bias, const float* colscale, const float* rowadd,
                 int rowadd_period, int accumulate, void* stream);

/* Fused-epilogue variants of nvit_gemm_nt (bf16 operands; N % 256 == 0, K % 64 == 0 — query with
 * nvit_gemm_nt_fusable; otherwise callers run nvit_gemm_nt followed by nvit_swiglu_fwd / nvit_qknorm_fwd).
 * nvit_gemm_nt_swiglu: uv[M,2F] = A B^T with B the perm=1 (u16|v16 interleaved) shadow of c_fc / proj, AND
 *   xm[M,F] = (gu*u)*silu(gv*v), gu/gv = gs[.]*gscale with gs = suv in the same interleaved order (NULL = 1)
 *   (model.py:148-154: the [M,8C] pre-activation is written once, never re-read in forward).
 * nvit_gemm_nt_qknorm: q/k/v projections (nparts stacked [C,K] weights starting at part part0: 0=q,1=k,2=v) with
 *   the per-head L2 normalise, sqk*c_q scale and [B,H,T,64] head split done on the fp32 accumulators
 *   (model.py:99-119); rq/rk [M,H] = 1/||.||.  Requires head dim 64 and n_embd % 256 == 0.
 * nvit_gemm_nt_swiglu_bwd: autograd of model.py:148-155 / 259-262 in one launch: dx[M,F] = A B^T (A = dL/dy of
 *   mlp_c_proj / out_proj, B = that weight's transposed shadow [F,K]) stays in the accumulators; with the saved raw
 *   uv[M,2F] (interleaved, as written by nvit_gemm_nt_swiglu) it writes duv[M,2F] (same layout) and, when gs != NULL
 *   (suv, natural order [u(F)|v(F)]), part[2*ceil(M/256), 2F] = per-128-row partial sums of d(suv) (natural order;
 *   reduce with nvit_colsum_reduce).  Replaces nvit_gemm_nt + nvit_swiglu_bwd.  Requires F % 256 == 0. */
/* Tile scheduling of the persistent NT GEMM kernels: 0 = static round-robin (default, or NVIT_GEMM_SCHED=static),
 * 1 = dynamic per-XCD ticket counters (NVIT_GEMM_SCHED=dynamic): robust when other kernels (RCCL) hold some CUs. */
int nvit_set_gemm_sched(int dynamic);
/* Kernel selection for tests / experiments (-1 = environment default).  nt_impl: 0 = 128x128 kernel only,
 * 1 = persistent 256-row kernels for large problems (default), 2 = persistent kernels whatever the tile count;
 * tn_impl: 0 = 128x128 kernel only, 1 = persistent 256x256 kernel for eligible shapes (default). */
int nvit_set_gemm_impl(int nt_impl, int tn_impl);
/* Variants of the persistent weight-gradient kernel (A/B measurements).  bit 0: 1 = XCD-contiguous work-item deal
 * (default: tiles that share operand panels run on one XCD and hit its L2), 0 = round-robin (NVIT_TN_ORDER=0);
 * +2 = ring of 4 x 32 KiB stages, +4 = ring of 2 x 64 KiB stages (default; NVIT_TN_RING=4 selects the former). */
int nvit_set_tn_order(int mode);
int nvit_gemm_nt_fusable(int dt, int M, int N, int K);
int nvit_gemm_nt_swiglu_bwd(int dt, const void* A, int lda, const void* B, int ldb, const void* uv, void* duv,
                            float* part, int M, int F, int K, const float* gs, float gscale, void* stream);
int nvit_gemm_nt_swiglu(int dt, const void* A, int lda, const void* B, int ldb, void* uv, void* xm, int M, int F,
                        int K, const float* gs, float gscale, void* stream);
/* q_prescale: extra factor folded into the q part (qh = q_prescale * sqk*c_q * unit vector, one rounding); pass the same
 * value to nvit_attn_fwd_bounded / nvit_attn_bwd_qknorm.  1 = plain. */
int nvit_gemm_nt_qknorm(int dt, const void* A, int lda, const void* B, int ldb, int M, int K, int nparts, int part0,
                        const float* sqk, float c_q, float q_prescale, void* qh, void* kh, void* vh, float* rq, float* rk,
                        int T, int H, int d, void* stream);

/* nvit_gemm_tn: weight gradient  G[N,K] (+)= sum_m A[m,N-col] * B[m,K-col]  over Mred rows.
 * A [Mred, N] (lda), B [Mred, K] (ldb) of type dt.  Split over `splits` row chunks into the
 * fp32 workspace ws [splits, N, K] (ws_bytes >= splits*N*K*4 + 256; the tail is no longer used - zero padding
 * comes from a device array owned by the library), then reduced in fixed order
 * (deterministic) into G (fp32, ld = ldg): G[perm(n)] = (accumulate ? G : 0) + sum.
 * perm as in nvit_shadow_weights (maps shadow row n to master row). */
int nvit_gemm_tn(int dt, const void* A, int lda, const void* B, int ldb, float* G, int ldg, int Mred, int N,
                 int K, int splits, float* ws, int64_t ws_bytes, int perm, int accumulate, void* stream);

/* ---- row-wise fused ops on the fp32 residual stream --------------------------------
 * nvit_lerp_fwd: out = nrm(nrm(h) + |alpha*c_a| * (nrm(y) - nrm(h)))   (model.py:134-142,159-167)
 * if skip_x != NULL additionally out = nrm(out*skip[0] + skip_x)         (norm_skip, model.py:84-87,452)
 * h, skip_x fp32 [M,C]; y of type y_dt; out fp32; out_lo (type dt, may be NULL) = cast(out). */
int nvit_lerp_fwd(int dt, const float* h, const void* y, int y_dt, const float* alpha, float c_a,
                  const float* skip_x, const float* skip, float* out, void* out_lo, int M, int C, void* stream);
/* nvit_lerp_bwd: given dout, recomputes the forward and writes dh (fp32; += if accum_dh), dy (fp32
 * and/or type-dt copy, either may be NULL), dskip_x (fp32, written, only if skip_x), and per-WAVE
 * partial sums part_dlam [4*nblk, C] (d/d|alpha*c_a|) and part_dskip [4*nblk] (if skip_x).
 * nblk = number of 4-wave workgroups the caller sizes the partial buffers for (<= 4096).
 * dout_add (bf16 [M,C] or NULL): the incoming gradient is dout + dout_add - lets the data-gradient GEMM that produced
 * dout_add store bf16 once instead of read-modify-writing the fp32 dout (the reference's autocast nn.Linear hands its
 * input gradient back in bf16 as well). */
/* nvit_lerp_bwd_blocks: the nblk that fills the device exactly once for the kernel variant these options select (0 on
 * error); callers size the partial buffers with it. */
int nvit_lerp_bwd_blocks(int dt, int y_dt, int C, int has_add, int has_skip, int accum);
int nvit_lerp_bwd(int dt, const float* dout, const void* dout_add, const float* h, const void* y, int y_dt, const float* alpha,
                  float c_a, const float* skip_x, const float* skip, float* dh, int accum_dh, float* dy,
                  void* dy_lo, float* dskip_x, float* part_dlam, float* part_dskip, int nblk, int M, int C,
                  void* stream);

/* RMSNorm (model.py:170-182; not on the nViT path, provided so that the public module works):
 * out = x * rsqrt(mean(x^2) + eps) * w, fp32 [M, C]; rstd [M] saved for the backward.
 * nvit_rmsnorm_bwd: dx [M, C] and part_dw [nblk, C] (per-workgroup partial sums of d(w); reduce with nvit_colsum_reduce). */
int nvit_rmsnorm_fwd(const float* x, const float* w, float eps, float* out, float* rstd, int M, int C, void* stream);
int nvit_rmsnorm_bwd(const float* dout, const float* x, const float* w, const float* rstd, float* dx, float* part_dw,
                     int nblk, int M, int C, void* stream);
/* Block.norm_skip on its own (model.py:84-87): out = nrm(src*skip[0] + tgt), fp32 [M,C]; backward writes dsrc, dtgt
 * (tgt == NULL / dtgt == NULL: the target term is absent, i.e. justnorm(src*skip[0]), model.py:43-44,89-90)
 * and part_dskip [nblk]. (ViT.forward uses the copy fused into nvit_lerp_fwd/bwd.) */
int nvit_norm_skip_fwd(const float* src, const float* tgt, const float* skip, float* out, int M, int C, void* stream);
int nvit_norm_skip_bwd(const float* dout, const float* src, const float* tgt, const float* skip, float* dsrc,
                       float* dtgt, float* part_dskip, int nblk, int M, int C, void* stream);

/* nvit_qknorm_fwd: per-head cosine normalise + learned scale + head split (model.py:104-119,231-247)
 * q/k/v sources: row-major, type dt, row stride ld* elements, C columns each.
 * qh = (sqk*c_q) * nrm_d(q) etc. written [B,H,T,d] type dt; v copied to [B,H,T,d];
 * rq, rk [M,H] fp32 = 1/||q_head||. */
int nvit_qknorm_fwd(int dt, const void* q, int ldq, const void* k, int ldk, const void* v, int ldv,
                    const float* sqk, float c_q, void* qh, void* kh, void* vh, float* rq, float* rk, int B,
                    int T, int H, int d, void* stream);
/* nvit_qknorm_bwd: dq/dk/dv (type dt, row strides ld*) from dqh/dkh/dvh [B,H,T,d]; part_dsqk
 * [nblk, C] partial sums of d/d(sqk*c_q). */
int nvit_qknorm_bwd(int dt, const void* dqh, const void* dkh, const void* dvh, const void* qh, const void* kh,
                    const float* rq, const float* rk, const float* sqk, float c_q, void* dq, int ldq,
                    void* dk, int ldk, void* dv, int ldv, float* part_dsqk, int nblk, int B, int T, int H,
                    int d, void* stream);

/* nvit_swiglu_fwd: x[m,j] = (gu*u) * silu(gv*v), u = uv[m, 32*(j/16) + j%16], v = uv[m, 32*(j/16)+16+j%16]
 * (interleaved GEMM output, see perm=1), gu = gscale*suv[j], gv = gscale*suv[F+j]; suv may be NULL (=1)
 * (model.py:148-154 with gscale = sqrt(C); cross-attention gate model.py:259-261 with suv=NULL). */
int nvit_swiglu_fwd(int dt, const void* uv, const float* suv, float gscale, void* x, int M, int F, void* stream);
/* nvit_swiglu_bwd: duv (interleaved, type dt) and part_dsuv [nblk_rows, 2F] (natural order, d/d(suv),
 * already multiplied by gscale; skipped if suv NULL). rows_per_blk rows per partial. */
int nvit_swiglu_bwd(int dt, const void* dx, const void* uv, const float* suv, float gscale, void* duv,
                    float* part_dsuv, int rows_per_blk, int M, int F, void* stream);

/* nvit_colsum_reduce: out[n] (+)= f(sum_b part[b, n]).  kind 0: plain*scale; kind 1: sum * sign(ref[n]*scale) * scale
 * (d|alpha c_a| -> d alpha); kind 2: plain*scale written to out[perm1(n)] (SwiGLU interleave -> natural order). */
int nvit_colsum_reduce(const float* part, int nblk, int N, float* out, int accumulate, int kind, const float* ref,
                       float scale, void* stream);
/* nvit_colsum_reduce_multi: up to 8 such reductions in ONE launch (host arrays of n entries; pointers as int64).  An item
 * may name a second partial array part_b (0 = none) whose reduced, scaled sum is added after the first one's. */
int nvit_colsum_reduce_multi(const int64_t* part, const int* nblk, const int64_t* part_b, const int* nblk_b, const int* N,
                             const int64_t* out, const int* accumulate, const int* kind, const int64_t* ref,
                             const float* scale, int n, void* stream);
/* nvit_colsum: out[n] (+)= scale * sum_r a[r,n] * (b ? b[r,n] : 1) over R rows; a,b fp32 or dt (a_dt,b_dt);
 * `period`>0 folds rows modulo period: out[(r%period), n] (pos-embed gradients). */
int nvit_colsum(const void* a, int a_dt, int lda, const void* b, int b_dt, int ldb, int R, int N, int period,
                float* out, int accumulate, float scale, void* stream);
int nvit_cast(const float* src, void* dst, int dt, int64_t n, void* stream);
/* Input pipeline, deterministic part (train.py:1084-1090): ToTensor + kornia Normalize(mean, std) in one pass.
 * in: uint8 [B,H,W,C] (in_is_u8_hwc = 1; scaled by 1/255 first) or fp32 [B,C,H,W] already in [0,1]; out fp32 [B,C,H,W]
 * = (in - mean) / std.  (The reference's randomised AutoAugment policy is third-party kornia code: not rebuilt.) */
int nvit_normalize_images(const void* in, int in_is_u8_hwc, float* out, int B, int C, int H, int W, float mean, float std_,
                          void* stream);
/* out[r,n] = a[r,n] * s[n] * c   (sz scaling model.py:466-468 and its backward) */
int nvit_scale_cols(const float* a, int lda, const float* s, float c, void* out, int out_dt, int ldo, int R, int N,
                    void* stream);

/* ---- attention ------------------------------------------------------------------------
 * O = softmax(scale * qh kh^T) vh per (b,h), non-causal, no mask (model.py:121-124 SDPA branch).
 * qh,kh,vh [B,H,T,d] type dt; o [B,T,H*d] type dt (heads merged, model.py:127); lse [B,H,T] fp32
 * (natural log of the softmax denominator, including the running max).  d in {32,64}.
 * impl: 0 = scalar-FMA reference kernel (any dt), 1 = MFMA flash kernel (bf16 only). */
int nvit_attn_fwd(int dt, int impl, const void* qh, const void* kh, const void* vh, float scale, void* o, float* lse,
                  int B, int H, int Tq, int Tk, int d, void* stream);
/* Same, for the nViT call sites where q and k are (sqk*c_q) * unit vectors per head (model.py:108-112): every score is
 * bounded by max_d (sqk_d*c_q)^2, and while that bound is small (it is 1 at initialisation) the MFMA kernel takes
 * probabilities relative to the bound instead of a running maximum (no per-tile max / rescale).  sqk: [H*d] fp32.
 * q_prescale > 0: qh holds q_prescale * q_hat (see nvit_gemm_nt_qknorm); the result is that of the un-scaled q.  With
 * q_prescale = scale * log2(e) the exponent of the MFMA kernel needs no multiply: the score accumulator starts at minus
 * the bound and goes straight into v_exp_f32 (one VALU instruction less per score in a VALU-bound kernel). */
int nvit_attn_fwd_bounded(int dt, int impl, const void* qh, const void* kh, const void* vh, float scale, const float* sqk,
                          float c_q, float q_prescale, void* o, float* lse, int B, int H, int Tq, int Tk, int d,
                          void* stream);
/* delta: [2,B,H,Tq] fp32 workspace (MINUS rowsum(dO*O) and MINUS lse in log2 units - accumulator seeds of the dk/dv
 * kernel). dqh,dkh,dvh [B,H,T,d] type dt. */
int nvit_attn_bwd(int dt, int impl, const void* dout, const void* qh, const void* kh, const void* vh, const void* o,
                  const float* lse, float scale, void* dqh, void* dkh, void* dvh, float* delta, int B, int H,
                  int Tq, int Tk, int d, void* stream);

/* nvit_attn_bwd_qknorm: MFMA attention backward (bf16, d=64) with nvit_qknorm_bwd fused into the epilogues:
 * writes token-major dq/dk/dv (type bf16, row stride ld elements, head h at column h*64) and the partial sums
 * part_q [B*ceil(Tq/128), C], part_k [B*ceil(Tk/128), C] of d/d(sqk*c_q) (reduce with nvit_colsum_reduce).
 * q_prescale: the factor the producer folded into qh (see nvit_attn_fwd_bounded); gradients are those of the plain form. */
int nvit_attn_bwd_qknorm(int dt, const void* dout, const void* qh, const void* kh, const void* vh, const void* o,
                         const float* lse, float scale, const float* rq, const float* rk, const float* sqk, float c_q,
                         float q_prescale, void* dq, int ldq, void* dk, void* dv, int ldkv, float* part_q, float* part_k,
                         float* delta, int B, int H, int Tq, int Tk, int d, void* stream);

/* ---- patch embedding / head / reconstruction -------------------------------------------
 * nvit_im2col: A_l [M, ch*Pl*Pl] and A_g [M, ch*Pg*Pg] (type dt, column order (c,ph,pw)) from
 * img fp32 [B,ch,S,S]; global windows are reflect-padded by (Pg-Pl)/2 and strided by Pl
 * (model.py:286-304,407-408).  The exact-fp32 mode multiplies these by the patch weights with nvit_gemm_nt; the
 * bf16 mode never materialises them (nvit_patch_embed_fwd). */
int nvit_im2col(int dt, const float* img, void* A_l, void* A_g, int B, int ch, int S, int Pl, int Pg, void* stream);
/* nvit_patch_embed_fwd: the dual patch embedding of the bf16 mode as ONE kernel (model.py:286-304 the two Conv2d
 * patchifiers, :407-415 their application + position embeddings):
 *   out_l[m] = W_l . patch_l(m) + b_l + pos_l[m % T],   out_g[m] = W_g . patch_g(m) + b_g + pos_g[m % T]   (fp32 [M, C])
 * Patches are gathered from img fp32 [B,ch,S,S] into LDS (same geometry rules as nvit_im2col), split into bf16
 * hi + lo on the fly and multiplied on the bf16 MFMA as hi*hi + lo*hi + hi*lo (fp32-accurate to ~2^-16 relative).
 * w_l / w_g: split weight images [C, 2*Kp] from nvit_shadow_weights (perm 2), Kp = nvit_patch_embed_kp(ch*P*P).
 * a_l / a_g: optional bf16 [ceil(M/256)*256, Kp] outputs = the rounded patch rows (columns >= K zero), the saved
 * operand of the weight-gradient GEMM; NULL to skip.  lo_l / lo_g: optional bf16 [M, C] twins of out_l / out_g (the
 * operands of the q and k/v projections; needs C % 8 == 0); NULL to skip.  b_l / b_g may be NULL. */
int nvit_patch_embed_kp(int K);
int nvit_patch_embed_fwd(const float* img, const void* w_l, const float* b_l, const float* pos_l, float* out_l, void* lo_l,
                         void* a_l, const void* w_g, const float* b_g, const float* pos_g, float* out_g, void* lo_g,
                         void* a_g, int B, int ch, int S, int Pl, int Pg, int C, void* stream);
/* mean over tokens + LayerNorm(eps) (model.py:455-456, mlp_head.0): x fp32 [B,T,C] ->
 * pooled [B,C] fp32, ln [B,C] fp32 and ln_lo (type dt, ld = C), stats [B,2] = {mean, rstd}. ws [B, nchunk, C]. */
int nvit_pool_ln_fwd(int dt, const float* x, const float* w, const float* b, float eps, float* pooled, float* ln,
                     void* ln_lo, float* stats, float* ws, int nchunk, int B, int T, int C, void* stream);
/* backward of the above: dln [B,C] fp32 -> dx [B,T,C] fp32 (written, broadcast /T), dw, db (+=). */
int nvit_pool_ln_bwd(const float* dln, const float* pooled, const float* w, const float* stats, float* dx,
                     float* dw, float* db, int accumulate, int B, int T, int C, void* stream);
/* recon loss (model.py:459-464): raw fp32 [M, ch*P*P] = x W_r^T + b_r; loss = mean((tanh(raw) - patch(img))^2).
 * part [nblk] partial sums; loss[0] written by a fixed-order final reduce. */
int nvit_recon_loss(const float* raw, const float* img, float* part, int nblk, float* loss, int B, int ch, int S,
                    int P, void* stream);

/* ---- Kohonen (SOM) head, BASELINE config C5 (kohonen.py:100-165, model.py:419-444,482-561) ---------
 * nvit_som_bmu: idx[m] = argmin_n ||x_m - node_n||_2 given scores[M,N] = x . node^T (nvit_gemm_nt, fp32) and the
 *   nodes [N,C] (kohonen.py:111-114 cdist+argmin); nn_ws [N] workspace; first minimum wins.
 * nvit_gather_rows / nvit_scatter_rows: repr = nodes[idx] (kohonen.py:117) and its backward (fixed order).
 * nvit_som_update: KohonenMap.update_nodes (kohonen.py:121-165) for the whole batch, in place on nodes [gm*gn, C]:
 *   B sequential steps; step i uses the BMU of FLAT token i and sample i mean-pooled T*C -> C,
 *   strength lr_alpha * exp(-d2 / (2 sigma^2)), d2 = squared grid distance, wrapped around the map edges when
 *   periodic != 0 (kohonen.py:80-98). v_ws [B,C], s_ws [B,gm*gn].
 * nvit_cos_consistency_*: 1 - mean_m cos(a_m, b_m) (model.py:482-491); stats [M,3] saved for backward.
 * nvit_huber_*: F.huber_loss(a, b) with delta 1, mean (model.py:441-442).
 * nvit_som_smooth_*: mean over tokens and 8 periodic grid neighbours of ||node[idx] - node[nb]|| for ONE map
 *   (model.py:503-561); cnt [N] int32 and D [N,8] workspaces are kept for the backward.
 * nvit_recon_bwd: d(raw) of mean((tanh(raw) - patch(img))^2) times g[0] (model.py:459-464), type dt. */
int nvit_som_bmu(const float* scores, const float* nodes, float* nn_ws, int64_t M, int N, int C, int64_t* idx,
                 void* stream);
int nvit_gather_rows(const float* nodes, const int64_t* idx, float* out, int64_t M, int C, void* stream);
int nvit_scatter_rows(const float* dout, const int64_t* idx, float* dnodes, int64_t M, int N, int C, void* stream);
/* out[M,N] fp32 one-hot rows of idx: the balanced form of the scatter is nvit_gemm_tn(onehot, dout) (exact). */
int nvit_onehot(const int64_t* idx, float* out, int64_t M, int N, void* stream);
int nvit_som_update(float* nodes, const float* x, const int64_t* idx, float lr_alpha, float sigma, int gm, int gn,
                    int periodic, float* v_ws, float* s_ws, int B, int T, int C, void* stream);
int nvit_cos_consistency_fwd(const float* a, const float* b, float* stats, float* part, int nblk, float* loss,
                             int64_t M, int C, void* stream);
int nvit_cos_consistency_bwd(const float* a, const float* b, const float* stats, const float* g, float* da, float* db,
                             int64_t M, int C, void* stream);
int nvit_huber_fwd(const float* a, const float* b, float* part, int nblk, float* loss, int64_t n, void* stream);
int nvit_huber_bwd(const float* a, const float* b, const float* g, float* da, float* db, int64_t n, void* stream);
int nvit_som_smooth_fwd(const float* nodes, const int64_t* idx, int* cnt, float* D, float* loss, int64_t M, int Nn,
                        int C, int map_size, void* stream);
int nvit_som_smooth_bwd(const float* nodes, const float* D, const int* cnt, const float* g, float* dnodes,
                        int accumulate, int64_t M, int Nn, int C, int map_size, void* stream);
int nvit_recon_bwd(int dt, const float* raw, const float* img, const float* g, void* draw, int B, int ch, int S, int P,
                   void* stream);

/* Loss side: F.cross_entropy(logits, Y) of train.py:906 with its gradient in the same pass.
 * rowloss[B] (workspace) = logsumexp(row) - row[label]; loss[0] = mean; dlogits[B,N] = (softmax - onehot) / B
 * (multiply by the upstream scalar gradient).  Labels outside [0,N) contribute 0 loss (and a plain softmax/B row). */
int nvit_ce_loss(const float* logits, const int64_t* labels, float* rowloss, float* loss, float* dlogits, int B, int N,
                 void* stream);

/* ---- gradient all-reduce by direct peer reads over xGMI (nvit/train.py:438-446) --------------
 * Symmetric flat fp32 buffers of n elements (n % 4 == 0), one per rank, mapped into this process (IPC); peer_ptrs is a
 * HOST array of nranks (<= 8) device pointers, [rank] = this rank's own buffer.  chunk = nvit_xgmi_chunk(n, nranks).
 * nvit_xgmi_reduce_scatter: own[rank*chunk ...] = scale * sum over ranks 0..nranks-1 (fixed order) of their chunk `rank`.
 * nvit_xgmi_all_gather   : own[j*chunk ...] = rank j's chunk j, for every j != rank.
 * The caller separates the phases (every rank finished writing / reduce-scatter / all-gather) with its own barrier. */
int64_t nvit_xgmi_chunk(int64_t n, int nranks);
int nvit_xgmi_reduce_scatter(const int64_t* peer_ptrs, int nranks, int rank, int64_t n, float scale, void* stream);
int nvit_xgmi_all_gather(const int64_t* peer_ptrs, int nranks, int rank, int64_t n, void* stream);

/* Device-synchronised form of the same collective (replaces the NCCL communicator the reference's DDP wrapper drives,
 * nvit/train.py:438-446, with phase flags that live on the devices): no host barrier between "gradients written",
 * reduce-scatter and all-gather, several regions ("slots", one per gradient bucket) in flight at once, every call just
 * kernel launches on `stream`.  Each rank owns a flag block in uncached device memory:
 *   nvit_xgmi_flag_bytes(nslots)            size of a flag block (0 for a bad slot count; at most 64 slots)
 *   nvit_xgmi_flags_alloc(nslots, &p, h)    allocate + zero this rank's block, h = 64-byte IPC handle to send to the peers
 *   nvit_xgmi_flags_open(h, &p) / _close(p) map / unmap a peer's block;  nvit_xgmi_flags_free(p) frees the own block
 *   nvit_xgmi_flags_error(own, nslots, &w, stream)   synchronises `stream`, w = 0 or (code << 8 | slot + 1) of the first
 *                                           wait that timed out ON ANY RANK: 1 reduce-scatter, 2 all-gather, 3 wait_gathered
 *   nvit_xgmi_set_timeout(seconds)          bound of every device-side wait (default 1800 s = the reference's 30-minute
 *                                           process-group timeout, train.py:224); set the same value on every rank
 *   nvit_xgmi_errword_alloc(&host, &dev) / _free(host)   one pinned host word the device can write
 * A timeout is fatal and fails closed: the timing-out rank stores the error word into EVERY rank's flag block, waits give
 * up as soon as their own error word is set, a kernel whose wait failed moves no data and announces no phase (nobody
 * gathers an unreduced chunk), and nvit_xgmi_wait_gathered copies the error word to the pinned host word it is given, so
 * the host can check it after every step without synchronising.
 * flag_ptrs: HOST array of nranks device pointers to the flag blocks ([rank] = own).  Region = elements [off, off + n) of
 * every symmetric buffer (multiples of 4), chunked by nvit_xgmi_chunk(n, nranks); `epoch` = number of this call for the
 * slot, from 1, the same on every rank.  A region may be rewritten only after nvit_xgmi_wait_gathered has been enqueued
 * for its slot (slots / epochs: HOST arrays of nwait <= 64 entries, passed to the kernel by value; host_err_dev: the
 * device pointer from nvit_xgmi_errword_alloc, or NULL) on the stream that rewrites it. */
int64_t nvit_xgmi_flag_bytes(int nslots);
int nvit_xgmi_flags_alloc(int nslots, void** dev_ptr, void* ipc_handle_out);
int nvit_xgmi_flags_open(const void* ipc_handle, void** dev_ptr);
int nvit_xgmi_flags_close(void* dev_ptr);
int nvit_xgmi_flags_free(void* dev_ptr);
int nvit_xgmi_flags_error(const void* own_flags, int nslots, unsigned* out, void* stream);
int nvit_xgmi_reduce_scatter_sync(const int64_t* peer_ptrs, const int64_t* flag_ptrs, int nranks, int rank, int nslots,
                                  int slot, unsigned epoch, int64_t off, int64_t n, float scale, void* stream);
int nvit_xgmi_all_gather_sync(const int64_t* peer_ptrs, const int64_t* flag_ptrs, int nranks, int rank, int nslots, int slot,
                              unsigned epoch, int64_t off, int64_t n, void* stream);
int nvit_xgmi_wait_gathered(const int64_t* flag_ptrs, int nranks, int rank, int nslots, const int* slots,
                            const unsigned* epochs, int nwait, void* host_err_dev, void* stream);
int nvit_xgmi_set_timeout(double seconds);
int nvit_xgmi_errword_alloc(void** host_ptr, void** dev_ptr);
int nvit_xgmi_errword_free(void* host_ptr);

/* Attention backward, dK/dV kernel: 1 = the hand-placed (generated-assembly) main loop where it applies (pre-scaled q,
 * the fused entry point; default, NVIT_ATTN_DKV_ASM=0 turns it off), 0 = the compiler-built kernel.  Both compute
 * bit-identical results (tests/test_gpu_ops.py). */
int nvit_set_attn_dkv_asm(int on);

#ifdef __cplusplus
}
#endif
#endif /* NVIT_HIP_H */
