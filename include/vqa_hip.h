/* C-ABI of libvqa_hip.so: the hand-written HIP (gfx950 / MI355X) kernels behind the drop-in VQA model.
 *
 * The reference (GalGoldstein/vqa) is pure Python and has no FFI of its own: its hot path is a chain of stock
 * torch.nn modules.  Each entry point below replaces the arithmetic of the torch call site(s) cited beside it
 * (file:line under the reference tree).  The Python binding is vqa_amd/_lib.py (ctypes); INTEGRATION.md shows it.
 *
 * Conventions
 *   - every function returns 0 on success, 1 = bad argument, 2 = launch failure (3 = over capacity, only
 *     vqa_soft_accuracy_sum_noalloc); nothing throws across the ABI;
 *   - `stream` is a hipStream_t (PyTorch's current stream); all work is asynchronous on it; the library keeps no
 *     state, caches no stream or device, allocates nothing, and is re-entrant (autograd worker threads call it) - the
 *     one exception is vqa_soft_accuracy_sum_noalloc, which owns a static per-GPU scratch array (see there);
 *   - the caller owns every buffer (inputs, outputs, workspaces, saved-for-backward tensors), all device memory;
 *   - speed-only switches are read from the environment once per process: VQA_PLAN_RESERVE,
 *     VQA_WALK_REV, VQA_NT_MIN_MB, VQA_CONV_PAIR, VQA_WGRAD_COF4 - they change the order in which a kernel walks its
 *     tiles, how many images share a tile, the size of a persistent grid or the cache policy of its accesses, never
 *     what is computed (outputs identical; partial-sum rows may be cut differently: fp32 summation order);
 *   - dtype codes: 0 = f32, 1 = bf16, 2 = f16 (image input only).  "T" below = the activation / MFMA dtype;
 *   - activations are NHWC; matrices are row-major with an explicit leading dimension in elements.
 */
#ifndef VQA_HIP_H
#define VQA_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

/* ---------------------------------------------------------------- image encoder (reference cnn.py:23-54) */

/* [B][C][H][W] images (f32/f16/bf16; dataset.py:96-98 stores f16) -> NHWC [B][H][W][CP] in T, channels >= C zero. */
int vqa_nchw_to_nhwc(void* stream, int in_dtype, int out_dtype, const void* in, void* out, int B, int C, int H,
                     int W, int CP);

/* torch-layout f32 conv weight [Cout][Cin][3][3] -> k-contiguous slab for vqa_conv3x3_fwd.
 * dgrad = 0: [CinP/CK][Cout][KP], k = tap*CK + c.   dgrad = 1: rotated 180 deg + transposed, for the data gradient. */
int vqa_conv_pack(void* stream, int dtype, const float* w, void* out, int Cout, int Cin, int CinP, int CK, int KP,
                  int dgrad);

/* The same for n (<= 24) weights in one launch: w / out are HOST arrays of device pointers, desc a host array of
 * n x 6 ints (Cout, Cin, CinP, CK, KP, dgrad). */
int vqa_conv_pack_all(void* stream, int dtype, int n, const void* const* w, void* const* out, const int* desc,
                      const void* const* scale /* NULL, or per job NULL / [Cout] f32 folded into a forward slab:
                                                  eval-mode BatchNorm, W' = W * gamma / sqrt(var + eps) */);

/* nn.Conv2d(k=3, stride 1, padding=pad) forward (cnn.py:23-40) as implicit GEMM on MFMA; also the data gradient
 * when called on dy with a dgrad slab and pad' = 2 - pad.  y = conv(x) + bias.  R x CW (<= 256 pixels) is the output
 * tile; G persistent workgroups per output-channel block walk the (image, tile) pairs - take G from
 * vqa_conv3x3_plan() (resident workgroups of the instantiation x 256 CUs; -1 = unsupported shape).  `stats` (may be
 * NULL) receives G rows of per-channel (sum, sum of squares) of y, [G][Cout][2], for the BatchNorm that follows.
 * x and y are addressed through 32-bit byte offsets (buffer resources): each must be smaller than 4 GiB.  The
 * multi-chunk kernel (Cin larger than one channel chunk: 32 for bf16, 16 for f32) further needs x < 2 GiB, Cin <= 256,
 * IH * IW and B * OH below 2^22 (signed 24-bit index products, float quotients) and `bias` 16-byte aligned; the BatchNorm statistics of a bf16
 * call are the sums over the ROUNDED outputs as stored (formed by MFMA from the tile on its way out).  Status 1 otherwise.
 * Two images per tile: a multi-chunk call with pad 0 whose map is one tile (R >= OH, CW >= OW), B even,
 * (2 OH + 2) * CW <= 256 and 2 IH * (CW + 2) <= 384 treats each PAIR of consecutive images as one image of 2 IH rows and
 * drops the two output rows that straddle them (the 10x10 map of cnn.py:40 fills 7 of a tile's 16 fragment slots alone);
 * vqa_conv3x3_plan() then returns G = B / 2 statistics rows.  Same outputs bit for bit; VQA_CONV_PAIR=0 turns it off.
 * in_scale / in_shift ([Cin] f32, both or neither; pad must be 0): x is the RAW output of the previous conv and the
 * input is relu(in_scale * x + in_shift) - that layer's nn.BatchNorm2d + nn.ReLU (cnn.py:23-24, ...), applied while
 * the input tile is staged, so the normalised activation is never written (bit-identical to vqa_bn_act_pool_fwd
 * with pool = 0 followed by this call without them).
 * bn_y / bn_coef (both or neither; data-gradient calls: bias NULL, relu 0, stats required): the output y = d is the
 * gradient w.r.t. the BatchNorm + ReLU output of the layer below (no pool in between), bn_y that layer's raw conv
 * output (same shape as y), bn_coef its [4][Cout] f32 scale, shift, mean, invstd; `stats` then receives that layer's
 * BatchNorm-backward partial sums (G rows of [Cout][2]: sum dz, sum dz*xhat, dz = d * 1[scale*bn_y + shift > 0]) -
 * what vqa_bn_bwd(apply = 0, pool = 0) would produce from (bn_y, d) in a pass of its own; feed vqa_bn_bwd_finalize.
 * relu: 0 = none, 1 = y = max(conv + bias, 0) (inference with the BatchNorm folded into w / bias: cnn.py:23-24 as one
 * kernel), 3 = that AND the nn.MaxPool2d(2, 2) behind it (cnn.py:27, :31): y is then [B][OH/2][OW/2][Cout] and the
 * full-resolution activation is never written.  relu = 3 needs bf16, OH, OW, R and CW even (CW == 16 for the
 * single-chunk kernel, Cin 16 or 32), no stats / in_scale / bn_y, and turns the two-images-per-tile form off; G from
 * vqa_conv3x3_plan(variant 4).  Same bits as relu = 1 followed by vqa_bn_act_pool_fwd with unit scale, zero shift and
 * pool = 1. */
int vqa_conv3x3_fwd(void* stream, int dtype, const void* x, const void* wpacked, void* y, const float* bias,
                    float* stats, int B, int IH, int IW, int Cin, int OH, int OW, int Cout, int pad, int R, int CW,
                    int G, int relu /* 1: y = max(conv + bias, 0) - eval mode with the BatchNorm folded into w, bias */,
                    const float* in_scale, const float* in_shift, const void* bn_y, const float* bn_coef);
int vqa_conv3x3_plan(int dtype, int B, int IH, int IW, int Cin, int OH, int OW, int Cout, int pad, int R, int CW,
                     int variant /* 0 plain, 1 with in_scale / in_shift, 2 with bn_y / bn_coef, 3 vqa_conv3x3_dgrad_wgrad */);
/* Both gradients of a pad-0 conv whose input is the BatchNorm + ReLU of the layer below (autograd of cnn.py:24-25, 28-29:
 * the second conv of a block) from ONE pass over dy and that layer's raw output: the data-gradient call of
 * vqa_conv3x3_fwd with bn_y / bn_coef (dy = this call's x with Cin channels, dx its y with Cout channels, pad = 2), which
 * also leaves the conv's weight gradient as G partial slabs wpart [G][Cin][9][Cout] f32 - x = relu(scale*bn_y + shift)
 * is formed in the epilogue from the values loaded for the sums, bit for bit the input the forward conv saw.  Sum the
 * slabs with vqa_conv3x3_wgrad_reduce_all (desc: G, Cin, Cout, real input channels of the conv).  bf16, Cin = Cout =
 * 16 or 32 (all channels in one workgroup); status 1 otherwise.  dx and the sums are bit-identical to vqa_conv3x3_fwd's;
 * the weight gradient is the same sum in another (fixed) order than vqa_conv3x3_wgrad's. */
int vqa_conv3x3_dgrad_wgrad(void* stream, int dtype, const void* dy, const void* wpacked, void* dx, float* stats,
                            float* wpart, int B, int IH, int IW, int Cin, int OH, int OW, int Cout, int pad, int R,
                            int CW, int G, const void* bn_y, const float* bn_coef);

/* First encoder layer, bf16 path: nn.Conv2d(3, 16, 3, padding=pad) (cnn.py:23) straight from the NCHW image
 * (in_dtype f16 / f32 / bf16; dataset.py:96-98 stores f16) - no NHWC copy, no channel padding.
 * w is the f32 master weight [16][3][3][3]; y is NHWC bf16 [B][OH][OW][16]; G persistent workgroups (<= tiles * B)
 * walk the R x CW output tiles, stats = [G][16][2] as in vqa_conv3x3_fwd.
 * Tile limits: R * CW <= 256; (R+2) * (CW+2) <= 511 for _fwd (wide tiles: its stores want long runs), <= 384 for
 * _wgrad.  Images < 2 GiB, y / dy < 4 GiB (32-bit buffer offsets).
 * _wgrad: dw [16][3][3][3] f32 from dy (NHWC bf16) and the image; part = [G][16][32] f32.
 * flip != 0 reads every image mirrored along its width: the whole-batch RandomHorizontalFlip of the training loop
 * (vqa_model.py:250, model.flip) without a flipped copy of the batch. */
int vqa_conv3x3_first_fwd(void* stream, int in_dtype, const void* images, const float* w, void* y, const float* bias,
                          float* stats, int B, int IH, int IW, int OH, int OW, int pad, int R, int CW, int G,
                          const float* wscale /* NULL or [16]: folded into w */, int relu, int flip);
int vqa_conv3x3_first_wgrad(void* stream, int in_dtype, const void* images, const void* dy, float* part, float* dw,
                            int G, int B, int IH, int IW, int OH, int OW, int pad, int R, int CW, int flip);
/* The same with the BatchNorm-backward apply pass of layer 1 (vqa_bn_bwd, apply = 1, pool = 0) folded in: y is the
 * layer's raw conv output, g the gradient w.r.t. its BatchNorm + ReLU output (both NHWC bf16 [B][OH][OW][16]) and
 * scale .. c2 the six [16] coefficient vectors of that pass; dy = scale * ((z > 0 ? g : 0) - c1 - xhat * c2) is never
 * stored (layer 1 has no data gradient: nobody else reads it) and never formed either: the kernel sums x (x) dz,
 * x (x) y and x over the pixels and combines the three with the coefficients at the end (dy is affine in dz and y).
 * part = [G][3][16][32] f32 here.  The three sums need only scale and shift (the ReLU mask), so the kernel can run on
 * a batch slice as soon as that slice's g exists - before c1 and c2, which need every slice, do: with dw == NULL the
 * slabs stay in `part` (mean, invstd, c1, c2 may then be NULL) and _bn_reduce, given the slabs of all slices as one
 * [G][3][16][32] array, sums them in fixed order and writes dw (autograd of cnn.py:23, as the calls above). */
int vqa_conv3x3_first_wgrad_bn(void* stream, int in_dtype, const void* images, const void* y, const void* g,
                               const float* scale, const float* shift, const float* mean, const float* invstd,
                               const float* c1, const float* c2, float* part, float* dw, int G, int B, int IH, int IW,
                               int OH, int OW, int pad, int R, int CW, int flip);
int vqa_conv3x3_first_wgrad_bn_reduce(void* stream, const float* part, int G, const float* scale, const float* mean,
                                      const float* invstd, const float* c1, const float* c2, float* dw);

/* Conv2d weight gradient (autograd of cnn.py:23-40): dw [Cout][Cin_real][3][3] f32 = sum over pixels of dy (x) x.
 * `part` is a [G][Cout][9][Cin] f32 workspace (one partial slab per workgroup group, summed in fixed order; with
 * dw == NULL the sum is left to vqa_conv3x3_wgrad_reduce_all).
 * in_scale / in_shift: as in vqa_conv3x3_fwd (x = the previous conv's raw output, pad == 0).
 * seam > 0: two images per tile, as vqa_conv3x3_fwd pairs them (a pad-0 map small enough for two in a tile) - here the
 * CALLER passes the pairs' geometry: B = images / 2, IH = 2 x the image's, OH = 2 x the real OH + 2, R >= OH (one tile
 * per pair), seam = the real OH; x and dy are the ordinary tensors (consecutive images are contiguous).  Same sums. */
int vqa_conv3x3_wgrad(void* stream, int dtype, const void* x, const void* dy, float* part, float* dw, int G, int B,
                      int IH, int IW, int Cin, int Cin_real, int OH, int OW, int Cout, int pad, int R, int CW,
                      const float* in_scale, const float* in_shift, int seam);
/* dw == NULL in vqa_conv3x3_wgrad: the partial slabs are left in `part` and this call reduces the slabs of n layers in
 * ONE launch (nine dependent little launches per training step otherwise).  part / dw: host arrays of n device
 * pointers; desc: host array of n x 4 ints (G, Cout, Cin, Cin_real).  Same sums in the same order per output. */
int vqa_conv3x3_wgrad_reduce_all(void* stream, int n, const void* const* part, void* const* dw, const int* desc);

/* nn.BatchNorm2d training statistics (cnn.py:23-40; eps 1e-5, momentum 0.1): from the conv partials to mean,
 * invstd, scale = gamma*invstd, shift = beta - mean*scale; running_mean / running_var updated in place
 * (unbiased variance); running_* may be NULL.  The `rows` partial rows are summed in fp64 in a fixed order. */
int vqa_bn_finalize(void* stream, const float* part, int rows, int C, double count, const float* gamma,
                    const float* beta, float* running_mean, float* running_var, float momentum, float eps,
                    float* mean, float* invstd, float* scale, float* shift,
                    long* num_batches_tracked /* NULL, or the layer's int64 counter: += 1 */);
/* eval-mode BatchNorm2d: scale / shift from the running statistics. */
int vqa_bn_eval_coeffs(void* stream, int C, const float* gamma, const float* beta, const float* running_mean,
                       const float* running_var, float eps, float* scale, float* shift);
/* The same for all n (<= 16) layers of the encoder in one launch (inference, cnn.py:23-40 under model.eval()): gamma ..
 * bias are HOST arrays of n device pointers (bias[i] = the conv's bias, may be NULL), C a host array of n channel counts;
 * coef [n][3][ldc] f32 receives scale, shift and the bias of the folded conv, bias * scale + shift. */
int vqa_bn_eval_coeffs_all(void* stream, int n, const void* const* gamma, const void* const* beta,
                           const void* const* running_mean, const void* const* running_var, const void* const* bias,
                           const int* C, float eps, float* coef, int ldc);
/* BatchNorm apply + nn.ReLU + nn.MaxPool2d(2,2) / nn.AvgPool2d(2,2) (cnn.py:23-41): pool 0 none, 1 max, 2 avg.
 * ysel (may be NULL; max pool only, same shape as a): receives the RAW conv output at the position the pool selects,
 * so that this layer's BatchNorm-backward sums can be formed at pooled resolution by the data gradient above it
 * (vqa_conv3x3_fwd with bn_y = ysel).  Status 1 when ysel is asked for a shape the row-staged kernel does not serve. */
int vqa_bn_act_pool_fwd(void* stream, int dtype, const void* y, void* a, const float* scale, const float* shift,
                        int B, int H, int W, int C, int pool, void* ysel);
/* Backward of pool + ReLU + BatchNorm (autograd of cnn.py:23-41).  apply = 0: partial sums of dz and dz*xhat into
 * `part` (vqa_bn_bwd_rows() rows of [C][2]); apply = 1: dy = scale*(dz - c1 - xhat*c2). */
int vqa_bn_bwd_rows(int dtype, int B, int H, int W, int C, int pool);
int vqa_bn_bwd(void* stream, int dtype, int apply, const void* y, const void* dout, void* dy, const float* scale,
               const float* shift, const float* mean, const float* invstd, const float* c1, const float* c2,
               float* part, int B, int H, int W, int C, int pool);
int vqa_bn_bwd_finalize(void* stream, const float* part, int rows, int C, double count, float* dgamma, float* dbeta,
                        float* c1, float* c2);

/* ---------------------------------------------------------------- linears (gru.py:51-52, vqa_model.py:59-103) */

/* MFMA GEMM with fused epilogue.  trans = 0: C[M,N] = A[M,K] * B[N,K]^T (nn.Linear forward, input gradients with a
 * pre-transposed weight).  trans = 1: C[M,N] = A[K,M]^T * B[K,N] (weight gradients).  Epilogue: + bias[n],
 * + rowadd[m / rowadd_group][n], activation (act = 1 ReLU, 3 SELU), C += (accumulate = 1), optional bf16 copy C2.
 * splits > 1: split-K over grid.z with raw partial sums in splitk_ws [splits][M][N] f32, summed in slice order.
 * emul [M][ldm] f32 / emode (NULL / 0: off): an elementwise factor applied last in the epilogue, for the head's
 * elementwise steps between its GEMMs (vqa_model.py:97-101).  emode 1: C2 = bf16(x * emul) while C = x - the Hadamard
 * product of the two projections as the bf16 copy of the second one (C2 required); emode 16 + act (act = 1 ReLU, 3
 * SELU): C = x * act'(emul), the derivative taken from the activation's OUTPUT emul - vqa_relu_bwd without a mask in
 * the GEMM that forms its incoming gradient; emode 32 + act (emul2 [M][ldm] and C2 required): C = x * emul2 *
 * act'(emul), C2 = bf16(x * emul * act'(emul2)) - vqa_hadamard_bwd without a mask (emul = fv, emul2 = fq).  Same
 * arithmetic, same bits as the separate kernels. */
int vqa_gemm(void* stream, int dtype, int trans, int M, int N, int K, const void* A, int lda, const void* B, int ldb,
             void* C, int ldc, int c_dtype, const float* bias, const float* rowadd, int rowadd_group, int act,
             int accumulate, void* C2, int ldc2, float* splitk_ws, int splits, const float* emul, int ldm, int emode,
             const float* emul2);
/* The classifier GEMM with the loss in its epilogue (vqa_model.py:103 self.fc(...) + :215 BCEWithLogitsLoss(sum)):
 * logits [M][ldc] f32 = A[M,K] * W[N,K]^T written once, loss[0] and dlogits [M][ldd] = sigmoid(logits) - target formed
 * from the accumulators (no second pass over logits / targets).  target: dense f32 [M][ldt]; or NULL with the sparse
 * form labels int64 / scores f32 [M][L] (as vqa_bce_sum_sparse); part: vqa_fc_bce_part(M, N) doubles of scratch
 * (per-workgroup partials, summed in a fixed order). */
int vqa_fc_bce_part(int M, int N);
int vqa_fc_bce(void* stream, int dtype, int M, int N, int K, const void* A, int lda, const void* W, int ldw,
               float* logits, int ldc, const float* target, int ldt, const long* labels, const float* scores, int L,
               float* dlogits, int ldd, double* part, float* loss);
/* out[0] = <a, b> (f32 data, f64 accumulation); part = 256 doubles of scratch. */
int vqa_dot(void* stream, const float* a, const float* b, long n, double* part, float* out);
/* out[n] = sum_m x[m][n] (bias gradients); `slabs` row slabs go through scratch [slabs][N] (NULL if slabs == 1). */
int vqa_colsum(void* stream, int dtype, const void* x, int M, int N, int ldx, float* out, float* scratch, int slabs);
/* weight_norm(dim=None) forward (vqa_model.py:59-75): W = v * g / sqrt(sumsq) as a T copy and a transposed T copy
 * (wt may be NULL). */
int vqa_wn_apply(void* stream, int dtype, const float* v, const float* g, const float* sumsq, int out, int in,
                 void* w, int ldw, void* wt, int ldt);
/* weight_norm backward: dv = (g/n)(dW - (dot/n^2) v), dg = dot/n with n = sqrt(sumsq), dot = <dW, v>. */
int vqa_wn_bwd(void* stream, const float* dw, const float* v, const float* g, const float* sumsq, const float* dot,
               long n, float* dv, float* dg);

/* The same for all n (<= 8) weight-normed layers of the head in two launches per direction.  Pointer / int arguments
 * are HOST arrays of length n holding device pointers; sumsq [n] f32 and part [n*256] f64 are device buffers.
 * fwd dims = n x 4 ints (out, in, ldw, ldt), wt[i] may be NULL; bwd dims = n x 2 ints (out, in). */
int vqa_wn_fwd_all(void* stream, int dtype, int n, const void* const* v, const void* const* g, void* const* w,
                   void* const* wt, const int* dims, float* sumsq, double* part);
int vqa_wn_bwd_all(void* stream, int n, const void* const* dw, const void* const* v, const void* const* g,
                   void* const* dv, void* const* dg, const int* dims, const float* sumsq, double* part);
/* dtype conversion with zero-padded leading dimension; scale (device scalar, may be NULL) multiplies every element (the
 * incoming gradient of a scalar loss, without a host read). */
int vqa_cast_pad(void* stream, int src_dtype, int dst_dtype, const void* src, int M, int N, int lds, void* dst,
                 int ldd, const float* scale);

/* dst [N][ldd] = src [M][lds]^T in dst_dtype, columns >= M zero: the transposed cast copy of an f32 master weight (the
 * input gradient of nn.GRU's first projection, gru.py:51-52, multiplies by W_ih^T). */
int vqa_cast_transpose(void* stream, int dst_dtype, const float* src, int M, int N, int lds, void* dst, int ldd);
/* nbytes (a multiple of 4, 4-byte aligned) of zeros: the fill the step needs before a scatter (the embedding gradient,
 * gru.py:48) without leaving the library. */
int vqa_zero(void* stream, void* p, long nbytes);

/* ---------------------------------------------------------------- question encoder (gru.py:48-52, :101-106) */

/* nn.Embedding gather (gru.py:103), time-major: x[t*B + b][0..ldx) = emb[q[b][t]], zero padded; q is int64 [B][T],
 * emb [V][E].  An id outside [0, V) (nn.Embedding raises on it) reads nothing: its row becomes NaN, which reaches the
 * loss; the backward skips it. */
int vqa_embed_fwd(void* stream, int dtype, const long* q, const float* emb, int B, int T, int E, int V, void* x,
                  int ldx);
/* its backward: demb[q[b][t]] += dx[t*B + b] (demb pre-zeroed by the caller), DETERMINISTIC: the tokens are sorted by
 * (id, position) and every row's occurrences are summed in that order (blocks of 16, then the blocks in order) - no
 * atomics, the same bits every run.  ws: vqa_embed_bwd_ws_bytes(B, T, E) bytes of scratch (-1 = unsupported size). */
int vqa_embed_bwd_ws_bytes(int B, int T, int E);
int vqa_embed_bwd(void* stream, const long* q, const float* dx, int B, int T, int E, int V, int lddx, float* demb,
                  void* ws);
/* nn.GRU cell gates (gru.py:104): r, z, n and h' = (1-z) n + z h from the two projections gi, gh ([B][3H], biases
 * included); save = [B][4H] (r, z, n, W_hn h + b_hn) for the backward.  H % 4 == 0 (16-byte accesses). */
int vqa_gru_gate_fwd(void* stream, int dtype, const float* gi, const float* gh, const float* hprev, int B, int H,
                     float* hnew, void* hnew_t, float* save);
/* dh_add (may be NULL): added to dhprev - the gradient a stacked layer above (nn.GRU num_layers > 1, gru.py:51) sends
 * into h_{t-1} through its input projection. */
int vqa_gru_gate_bwd(void* stream, int dtype, const float* dh, const float* save, const float* hprev, int B, int H,
                     void* dgi, void* dgh, float* dhprev, const float* dh_add);

/* The recurrence of nn.GRU (gru.py:104) in its per-step form, all T steps issued by one call (the step is host-bound:
 * one binding call instead of 2 T): step t = vqa_gemm(h_t[t] x W_hh^T + b_hh -> gh) then vqa_gru_gate_fwd.  gi
 * [T][B][3H] f32 (input projections, b_ih included), w_hh T [3H][ldw], h_f f32 / h_t T [T+1][B][H] with slot 0 = h_-1
 * = 0 filled by the caller, save [T][B][4H], gh [B][3H] f32 scratch, splitk_ws [splits][B][3H] f32 (NULL if splits 1). */
int vqa_gru_fwd_steps(void* stream, int dtype, const float* gi, const void* w_hh, int ldw, const float* b_hh,
                      float* h_f, void* h_t, float* save, float* gh, float* splitk_ws, int splits, int B, int T, int H);
/* BPTT of the same: step t = vqa_gru_gate_bwd then dh_{t-1} += dgh_t x W_hh (vqa_gemm, accumulate).  w_hh_t T
 * [H][ldw] (W_hh^T, 3H contiguous), outputs dgi / dgh T [T][B][3H], dh_ws [2][B][H] f32 scratch, splitk_ws
 * [splits][B][H] f32.  dh_seq (NULL for the top layer): f32 [T][B][H], the gradient a stacked layer above sends into
 * every step's output; dh_last must already include dh_seq[T-1]. */
int vqa_gru_bwd_steps(void* stream, int dtype, const float* dh_last, const float* save, const float* h_f,
                      const void* w_hh_t, int ldw, void* dgi, void* dgh, float* dh_ws, float* splitk_ws, int splits,
                      int B, int T, int H, const float* dh_seq);

/* The same recurrence (gru.py:104) with ONE launch per time step (csrc/gru_step.hip): the recurrent GEMM streams both
 * operands, kept in MFMA-fragment order in memory, straight into registers (no LDS staging, no split-K pass) and the
 * gate arithmetic runs in its epilogue on the accumulators - gh (forward) and the dh round trip (backward) never
 * exist in memory.  bf16 only, H % 64 == 0; status 1 = shape not served (nothing launched): use the *_steps form.
 *   vqa_gru_pack_whh   f32 master W_hh [3H][H] -> wf (forward operand) and / or wb (W_hh^T, backward operand), 3 H H
 *                      bf16 each, either may be NULL; once per training step, replaces the cast and the transposed copy
 *   vqa_gru_fwd_fused  as vqa_gru_fwd_steps; h_p = scratch of (T+1) x P(H) bf16, P(k) = ceil(B/64) * 64 * k: the
 *                      packed h_t each step writes for the next one (slot 0 is never read)
 *   vqa_gru_bwd_fused  as vqa_gru_bwd_steps; dgh_p = scratch of 2 x P(3H) bf16 */
int vqa_gru_pack_whh(void* stream, const float* w_hh, int H, void* wf, void* wb);
/* (vqa_gru_fwd_fused writes slot 0 of h_f / h_t itself - zeros - in its first launch: the caller need not fill it) */
int vqa_gru_fwd_fused(void* stream, const float* gi, const void* wf, const float* b_hh, float* h_f, void* h_t, void* h_p,
                      float* save, int B, int T, int H);
int vqa_gru_bwd_fused(void* stream, const float* dh_last, const float* save, const float* h_f, const void* wb, void* dgi,
                      void* dgh, void* dgh_p, float* dh_ws, int B, int T, int H, const float* dh_seq);

/* The whole nn.GRU recurrence (gru.py:104) in ONE persistent launch: each workgroup keeps its 48 x H slice of W_hh
 * (bf16) in LDS for all T steps, gates fused in registers, one device-wide barrier per step.  The grid is sized to be
 * co-resident (occupancy query x CUs; max_wg > 0 caps it lower so the recurrence holds fewer CUs beside other work).
 * The barrier spins are bounded: on expiry flags[1] is set and h_T (forward) / the gate gradients (backward) are
 * written as NaN - a timeout is visible in the loss, never a silently wrong number.  bf16 only, H % 256 == 0,
 * 48*(2H+32) <= 160 KiB; otherwise returns 1 and the caller uses vqa_gemm + vqa_gru_gate_fwd per step.
 * gi [T][B][3H] f32, h_f / h_t [T+1][B][H] (slot 0 = zeros), save [T][B][4H], flags = 2 unsigned of scratch. */
int vqa_gru_fwd_persistent(void* stream, const float* gi, const void* w, int ldw, const float* b_hh, float* h_f,
                           void* h_t, float* save, int B, int T, int H, unsigned* flags, int max_wg);
/* BPTT of the same recurrence in one launch: wt = W_hh^T bf16 [H][ldw]; outputs dgi, dgh bf16 [T][B][3H]. */
int vqa_gru_bwd_persistent(void* stream, const void* wt, int ldw, const float* h_f, const float* save,
                           const float* dh_last, void* dgi, void* dgh, int B, int T, int H, unsigned* flags,
                           int max_wg);

/* ---------------------------------------------------------------- attention / fusion head (vqa_model.py:83-103) */

/* vqa_model.py:90-93: s = a . w_s per region, softmax over the K regions, vhat = sum_k p_k v[b,k,:].
 * 16-byte accesses: H % 8 == 0 (bf16) / H % 4 == 0 (f32), D % 4 == 0, D <= 1024. */
int vqa_attn_fwd(void* stream, int dtype, const void* a, const void* ws, const float* v, int B, int K, int H, int D,
                 float* p, float* vhat, void* vhat_t);
int vqa_attn_bwd(void* stream, int dtype, const void* a, const void* ws, const float* v, const float* p,
                 const float* dvhat, int B, int K, int H, int D, float* dv, void* dpre, float* dws_part, int act);
/* out[b] = sum_k x[b*K + k] (gradient of the question term broadcast over regions, vqa_model.py:83-87). */
int vqa_group_sum(void* stream, int dtype, const void* x, int B, int K, int H, float* out, void* out_t);
/* vqa_model.py:99-101: fv * fq (* dropout mask), and its backward through both activations.
 * act (here and in vqa_attn_bwd / vqa_gemm): 1 = nn.ReLU, 3 = nn.SELU (vqa_model.py:54; the backward forms the
 * derivative from the activation's saved output).  mask / add: out = x * mask + add, either may be NULL - nn.Dropout
 * is mask = keep / (1-p); nn.AlphaDropout (vqa_model.py:55, selu branch) is mask = a * keep, add = a * alpha' *
 * (1 - keep) + b with torch's a, b, alpha' for p = 0.5. */
int vqa_hadamard_fwd(void* stream, int dtype, const float* fv, const float* fq, const float* mask, long n, void* out,
                     const float* add);
int vqa_hadamard_bwd(void* stream, int dtype, const float* dm0, const float* fv, const float* fq, const float* mask,
                     long n, void* dpv, void* dpq, int act);
int vqa_relu_bwd(void* stream, int dtype, const float* g, const float* y, const float* mask, long n, void* out, int act);
int vqa_mask_mul(void* stream, int dtype, const float* x, const float* mask, long n, void* out, const float* add);

/* ---------------------------------------------------------------- loss, accuracy, optimizer */

/* nn.BCEWithLogitsLoss(reduction='sum') (vqa_model.py:215) and its gradient sigmoid(x) - t in one pass. */
int vqa_bce_sum(void* stream, const float* logits, const float* target, int B, int C, int ldx, float* dlogits,
                double* part, float* loss);
/* vqa_model.py:127-130 without the per-sample host loop: acc[0] = sum_b target[b][argmax_c logits[b][c]]; the arg-max
 * is torch.argmax's (first maximum, NaN largest), always a valid index.  rows: B floats of scratch. */
int vqa_soft_accuracy_sum(void* stream, const float* logits, const float* target, int B, int C, int ldx, float* rows,
                          float* acc);

/* NO COUNTERPART IN THE REFERENCE.  vqa_soft_accuracy_sum for callers that cannot allocate or hand over scratch: code
 * under HIP graph capture, tight serving / evaluation loops, harnesses that forbid hipMalloc.  Same result bit for bit;
 * the B floats of scratch live in a file-scope device array of the library (256 KiB per GPU, held until the process
 * ends; its address is looked up once per device, on the first call there - hipGetSymbolAddress is not a stream
 * operation, so make that first call OUTSIDE a graph capture; every later call is two kernel launches and nothing
 * else).  No hipMalloc / hipFree, no workspace argument.  B <= 65536: a larger batch returns 3 (VQA_ERR_CAPACITY) and launches nothing.  All calls on one GPU share
 * that GPU's array: calls that overlap there (two streams, two threads) race; calls ordered on one stream do not. */
int vqa_soft_accuracy_sum_noalloc(void* stream, const float* logits, const float* target, int B, int C, int ldx,
                                  float* acc);

/* The same two from SPARSE soft-score targets: the (labels, scores) lists that dataset.py:135-138
 * scatters into the dense [B][C] matrix.  labels int64 [B][L] (< 0 = empty slot, distinct within a row), scores f32
 * [B][L], L <= 256.  rowloss: B doubles of scratch; rows: B floats of scratch. */
int vqa_bce_sum_sparse(void* stream, const float* logits, const long* labels, const float* scores, int B, int C, int L,
                       int ldx, float* dlogits, double* rowloss, float* loss);
int vqa_soft_accuracy_sum_sparse(void* stream, const float* logits, const long* labels, const float* scores, int B,
                                 int C, int L, int ldx, float* rows, float* acc);
/* nn.utils.clip_grad_norm_(max_norm, 2) + optim.Adamax step (vqa_model.py:259, :218, :266) over one flat f32
 * parameter / gradient buffer, no host synchronisation; max_norm <= 0 skips the clip.  part = 1024 doubles of scratch. */
int vqa_clip_adamax(void* stream, float* p, const float* g, float* m, float* u, long n, float max_norm, float lr,
                    float b1, float b2, float eps, int step, double* part, float* sq, float* coef,
                    float* total_norm);

/* ---------------------------------------------------------------- one call for a recorded sequence of calls
 * NO COUNTERPART IN THE REFERENCE (its layers are one Python call each: cnn.py:49-54 `self.convolutions(x)`).
 * vqa_run_plan issues, in order, a sequence of the stream-taking entry points above that vqa_amd/plan.py recorded from
 * one ordinary pass: words = [magic, n_ops] + per op [entry (vqa_plan_opcode(name), or -1 = "stream s waits for what
 * stream t has been given so far"), stream index, n slots] + per slot [kind, value]: 0 int, 1 float bits, 2 double bits,
 * 3 pointer as (base index << 48 | byte offset), 4 NULL, 5 / 6 host array of n pointers / ints (n inline words follow).
 * bases = this call's base addresses (scratch arena, inputs, parameters, gradient buffer ...), streams = hipStream_t
 * handles (index 0 = the caller's current stream).  Stateless: nothing is kept between calls, no allocation; an
 * ordering point records and waits for the caller's k-th event, or - events == NULL - creates, uses and destroys one.  Returns 0, or the failing op's status |
 * (1-based op index << 8); a malformed plan is status 1. */
int vqa_plan_opcode(const char* name);
int vqa_run_plan(const long* words, long n_words, void* const* bases, int n_bases, void* const* streams, int n_streams,
                 void* const* events /* NULL, or one hipEvent_t per ordering point, in order */, int n_events);

/* ---------------------------------------------------------------- hardware-layout probe (tests only) */
int vqa_probe_tr16(void* stream, const void* in, float* out);

#ifdef __cplusplus
}
#endif
#endif /* VQA_HIP_H */
