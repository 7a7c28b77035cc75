/*
 * mustafar_hip.h -- C ABI of libmustafar_hip.so (MI355X / gfx950).
 *
 * Drop-in boundary of the Mustafar sparse-attention decode path.  Every entry point is
 * `extern "C"`, takes plain device pointers / sizes and a `hipStream_t` passed as `void*`,
 * launches asynchronously on that stream and returns a hipError_t value as `int`
 * (0 == hipSuccess; MUSTAFAR_EINVAL for arguments the kernels do not support).
 * No torch types cross this boundary.  Citations are relative to the reference tree.
 *
 * Compressed-cache format (kernel/compression.py:8-247, consumed by kernel/csrc/SpMM_Kernel.cuh):
 *   bmp        u64 [B', tiles]       bit (63-i) set <=> element i of the 64-element tile != 0
 *   idx        u32 [B', tiles+1]     exclusive prefix of ceil8(nnz)/2 per tile (half2 units)
 *   NZ         fp16 stream           per tile ceil8(nnz) halfs: non-zeros in element order, then 0 padding
 *   NZ_offset  u32 [B']              start of each head's stream in uint4 (16-byte) units
 *   K tile id = (token/64)*D + d            -> 64 consecutive tokens of channel d
 *   V tile id = (token/64)*D + (c/64)*64 + token%64 -> 64 consecutive channels of one token
 *   B' = batch * kv_heads, tiles = T*D/64, D = head_dim (this build: D == 128).
 */
#ifndef MUSTAFAR_HIP_H
#define MUSTAFAR_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MUSTAFAR_EINVAL 1 /* == hipErrorInvalidValue */

/* Version / build probe: returns MAJOR*100 + MINOR. */
int mustafar_abi_version(void);

/*
 * Key SpMV.  Replaces `Key_SplitK_API` (kernel/build/SpMM_API.cuh:46-64; kernel/csrc/SpMM_API.cu:86-139),
 * the function `mustafar_key_formulation` calls (kernel/kernel_wrapper/mustafar_wrapper.cu:113-130).
 *   C[b, n, m] = fp16( sum_k fp32(Khat_g[m, k]) * fp32(B[b, n, k]) ),  g = b / num_key_value_groups
 *   A           unused (NULL in the reference too)
 *   B           fp16 [Batch_Size, N_Global, K_Global]   (query rows; the reference pads to N_Global = 8)
 *   C           fp16 [Batch_Size, N_Global, M_Global]   every element is written (no pre-zeroing needed)
 *   M_Global    = compressed tokens T (multiple of 64), K_Global = head_dim (128)
 *   N_Global    1 or 8 (the reference supports only 8; 1 is the un-padded fast form)
 *   Reduction_Workspace, Split_K   accepted for signature parity; unused (Split_K must be 1)
 */
int Key_SplitK_API(void* stream, const void* A, const uint64_t* bmp, const void* NZ, const uint32_t* idx,
                   const uint32_t* NZ_offset, const void* B, void* C, int M_Global, int N_Global, int K_Global,
                   void* Reduction_Workspace, int Split_K, int Batch_Size, int num_key_value_groups);

/*
 * Value SpMV.  Replaces `Value_SplitK_API` (kernel/build/SpMM_API.cuh:92-110; SpMM_API.cu:193-254),
 * called by `mustafar_value_formulation` (mustafar_wrapper.cu:242-260).
 *   C[b, n, m] = fp16( sum_k fp32(Vhat_g[k, m]) * fp32(B[b, n, k]) )
 *   B           fp16 [Batch_Size, N_Global, K_Global]   (softmax probabilities, K_Global = T)
 *   C           fp16 [Batch_Size, N_Global, M_Global]   M_Global = head_dim (128)
 *   Split_K     number of token chunks processed by independent workgroups (>= 1).  The reference
 *               carries the parameter but pins it to 1 (mustafar_wrapper.cu:199); here it is live:
 *               use mustafar_value_pick_split_k().  With Split_K > 1, Reduction_Workspace must hold
 *               mustafar_value_workspace_bytes() bytes (fp32 partials + flags; the reference's
 *               SplitK_Reduction, Reduction_Kernel.cuh:26-48, used fp16 partials).
 */
int Value_SplitK_API(void* stream, const void* A, const uint64_t* bmp, const void* NZ, const uint32_t* idx,
                     const uint32_t* NZ_offset, const void* B, void* C, int M_Global, int N_Global, int K_Global,
                     void* Reduction_Workspace, int Split_K, int Batch_Size, int num_key_value_groups);

int     mustafar_value_pick_split_k(int M_Global, int N_Global, int K_Global, int Batch_Size, int num_key_value_groups);
int64_t mustafar_value_workspace_bytes(int M_Global, int N_Global, int K_Global, int Batch_Size,
                                       int num_key_value_groups, int Split_K);

/*
 * Magnitude prune.  Replaces `dh_prune_key` / `dh_prune_value` (models/llama_mustafar_kernel.py:77-153):
 * per row of D halfs, thr = kth smallest |x| (1-indexed), out = |x| >= thr ? x : x*0 (sign-preserving zero).
 * `kth` = max(1, int(target_sparsity * D)) is computed by the caller (model :97).  In-place allowed.
 */
int mustafar_prune_magnitude(void* stream, const void* x, void* out, int64_t n_rows, int D, int kth);

/*
 * Compression.  Replaces `convert_key_batched` / `convert_value_batched` (kernel/compression.py:249-432)
 * as two device passes (each ONE read of x) with one small device->host read between them (the reference needs >= 1 + 2B'):
 *   pass 1  mustafar_compress_bitmap_{key,value}: x fp16 [B', t, D] (already pruned) ->
 *           bmp i64 [B', t*D/64], accum i32 [B', t*D/64 + 1] (accum_counts, compression.py:294-298) and
 *           head_off i64 [B'+1] = exclusive prefix of 2*accum[h][-1] (start of each head in halfs; :302-304).
 *   pass 2  mustafar_compress_pack_{key,value}: writes every half of nz_flat[0 .. head_off[B'])
 *           (non-zeros and zero padding; no pre-zeroing needed); packs x by the bitmaps of pass 1.
 */
int mustafar_compress_bitmap_key(void* stream, const void* x, int Bp, int t, int D, int64_t* bmp, int32_t* accum,
                                 int64_t* head_off);
int mustafar_compress_bitmap_value(void* stream, const void* x, int Bp, int t, int D, int64_t* bmp, int32_t* accum,
                                   int64_t* head_off);
/* Pass 1 with a second copy of head_off for a host that WAITS for it (round 6; kernel/compression.py:308 `total_size = ....item()`: the reference
 * synchronises here too).  host_mirror: B' + 1 int64 of device-visible host memory (hipHostMalloc / a pinned torch tensor) the caller has filled
 * with a negative sentinel; the launch that computes head_off stores every entry there as well, each one aligned 8-byte system-scope store,
 * BEFORE the launch that completes the offsets inside the blocks -- the caller polls until no entry is negative (or the stream has drained)
 * and sizes the packed buffer ~15 us earlier than a device-to-host copy behind the stream would let it.  head_off (device) is written as by
 * the plain calls: pass 2 reads that one.  key != 0: K tile geometry.  host_mirror == NULL: exactly mustafar_compress_bitmap_{key,value}. */
int mustafar_compress_bitmap_mirrored(void* stream, const void* x, int Bp, int t, int D, int key, int64_t* bmp, int32_t* accum,
                                      int64_t* head_off, int64_t* host_mirror);
int mustafar_compress_pack_key(void* stream, const void* x, int Bp, int t, int D, const int64_t* bmp,
                               const int32_t* accum, const int64_t* head_off, void* nz_flat);
int mustafar_compress_pack_value(void* stream, const void* x, int Bp, int t, int D, const int64_t* bmp,
                                 const int32_t* accum, const int64_t* head_off, void* nz_flat);

/*
 * Fused decode attention of one layer (extension).  Replaces the PyTorch glue of the reference hook
 * between the two SpMVs (models/llama_mustafar_kernel.py:270-317): window append, q.K_window^T, concat, / sqrt(d),
 * fp32 softmax, p.V_window, sum.  TWO launches on `stream` by default (T > 0): the one-pass launch -- every 64-token block runs key phase ->
 * softmax step -> value phase, the dense window rides along as a few more workgroups, one slab (max, sum, unnormalised output) per
 * workgroup and head -- and the row kernel that merges the slabs; MUSTAFAR_FLAG_TWO_LAUNCH (and T == 0) select the round-1 structure
 * of four launches: key SpMV (+ window scores) -> softmax -> value SpMV (+ window p.V partials) -> sum of the partials.
 * All operands fp16 unless noted; B' = Batch_Size / num_key_value_groups.
 *   q            [Batch_Size, 128]
 *   k_window     [B', window_capacity, 128]  rows [0, window_len) valid after the call; k_new [B', 128] (or NULL if the
 *   v_window     same                        newest row is already stored) is appended at row window_len - 1
 *   scores       [Batch_Size, ld_scores] scratch (scores, then probabilities); ld_scores >= T + window_len, % 8 == 0
 *   out          [Batch_Size, 128]
 *   workspace    mustafar_decode_workspace_bytes() bytes (fp32 partial slabs); Split_K as for Value_SplitK_API
 *   T            compressed tokens (multiple of 64, 0 allowed: window only; rows longer than 32768 take a streaming
 *                softmax); sqrt_d = sqrt(head_dim)
 *   window_len_extra  NULL, or a device int added to window_len inside the kernels (clamped to the capacity):
 *                lets a captured hipGraph of a whole decode step be replayed while the windows keep growing;
 *                advance it once per step with mustafar_counter_add().  ld_scores must then cover the capacity.
 *   attention_mask    NULL, or the hook's additive fp16 mask (models/llama_mustafar_kernel.py:293-301; the non-flash model
 *                always passes one, :723-728): rows of >= T + window length halfs, `mask_row_stride` halfs apart, one row
 *                per `heads_per_mask_row` consecutive score rows (the reference mask is [bsz, 1, 1, kv_len]: stride kv_len,
 *                heads_per_mask_row = q heads).  Applied exactly as the hook does: fp16(score / sqrt(d)) + mask in fp16,
 *                max with finfo(fp16).min, then the fp32 softmax.  Rows need no alignment.
 *   flags        0, or this call's own choice of FMA engine and launch structure (MUSTAFAR_FLAG_*), which then overrides the
 *                process defaults (environment / mustafar_set_fma_engine / mustafar_set_onepass) for this call only: two hook
 *                instances in one process can run different engines.  Undefined bits are rejected.
 */
#define MUSTAFAR_FLAG_ENGINE_FMA_MIX 1u  /* v_fma_mix_f32 per tile and head (exact products, fp16 subnormals included) */
#define MUSTAFAR_FLAG_ENGINE_MFMA    2u  /* v_mfma_f32_4x4x4_16B_f16 as a 4-wide FMA unit (GQA-4) */
#define MUSTAFAR_FLAG_ENGINE_DOT2    3u  /* v_dot2_f32_f16 on pairs of tiles (GQA-4 one-pass form; the default) */
#define MUSTAFAR_FLAG_TWO_LAUNCH     (1u << 4)  /* key SpMV -> softmax rows -> value SpMV -> sum */
#define MUSTAFAR_FLAG_ONE_PASS       (2u << 4)  /* one launch + slab merge */
int mustafar_decode_attention(void* stream, const uint64_t* k_bmp, const void* k_nz, const uint32_t* k_idx,
                              const uint32_t* k_nz_offset, const uint64_t* v_bmp, const void* v_nz, const uint32_t* v_idx,
                              const uint32_t* v_nz_offset, const void* q, void* k_window, void* v_window, const void* k_new,
                              const void* v_new, int window_len, int window_capacity, void* scores, int ld_scores, void* out,
                              void* workspace, int Split_K, int T, int Batch_Size, int num_key_value_groups, float sqrt_d,
                              const int32_t* window_len_extra, const void* attention_mask, int64_t mask_row_stride,
                              int heads_per_mask_row, uint32_t flags);
int64_t mustafar_decode_workspace_bytes(int T, int Batch_Size, int num_key_value_groups, int Split_K);

/*
 * Compressed cache with spare capacity (extension; SURVEY 8f rank 2: the reference re-copies bitmaps, offsets and every
 * head's stream on each 256-token append, models/llama_mustafar_kernel.py:339-390).  Same four arrays as the reference
 * format, but a head's bitmap / offset rows may be longer than the tokens in use, and NZ_offset may leave room behind
 * every head's stream -- so an append writes only the new tokens:
 *   bmp  [B', bmp_head_stride]   tiles [0, 2T) of a row in use        (bmp_head_stride >= 2T;     0 = exactly 2T)
 *   idx  [B', idx_head_stride]   entries [0, 2T] of a row in use      (idx_head_stride >= 2T + 1; 0 = exactly 2T + 1)
 */
typedef struct mustafar_cache_view {
    uint64_t* bmp;
    void*     nz;
    uint32_t* idx;
    uint32_t* nz_offset;
    int64_t   bmp_head_stride;
    int64_t   idx_head_stride;
    int64_t   nz_head_stride;   /* != 0: nz_offset[h] == h * nz_head_stride (uint4 units, < 2^32): the SpMV kernels then compute
                                   the head's stream start instead of loading it (one memory latency less per wave) */
} mustafar_cache_view;

/*
 * mustafar_decode_attention over a cache that GROWS BY EXTENTS (the 256-token trigger of models/llama_mustafar_kernel.py:324-398
 * without copying or moving what is already compressed): tokens [0, T_base) live in the two base views, tokens
 * [T_base + 256 i, T_base + 256 (i + 1)) in entry i of `k_extents` / `v_extents` -- arrays of mustafar_cache_view in DEVICE memory,
 * each describing a 256-token cache of its own (offsets relative to that extent; nz_head_stride != 0).  An entry is written once,
 * before the first call that names it, and never changed: a captured hipGraph of a call stays valid while the cache grows
 * behind it.  T_base and T - T_base are multiples of 256.  Read by the pair form of the one-pass launch (every group count: GQA-4 on the three engines,
 * GQA-2 and MHA on v_fma_mix; ld_scores % 32 == 0; mustafar_decode_reads_extents() tells); MUSTAFAR_EINVAL otherwise -- consolidate into one view then.
 *   T_device   NULL: T is the number of compressed tokens.  Otherwise a device int holding the compressed tokens IN USE
 *              (T_base <= *T_device <= T, a multiple of 256 beyond T_base), and T is the CAPACITY the launch is sized for: grid,
 *              slabs, score scratch (ld_scores >= T + window capacity), mask columns.  The caller adds 256 to *T_device when a
 *              trigger has listed an extent (and takes 256 off its window_len_extra counter, the windows having slid): ONE
 *              captured graph of the step then serves every cache length up to T (tests/test_gpu_extents.py).
 */
int mustafar_decode_attention_extents(void* stream, const mustafar_cache_view* k_base, const mustafar_cache_view* v_base, int T_base,
                                      const mustafar_cache_view* k_extents, const mustafar_cache_view* v_extents,
                                      const void* q, void* k_window, void* v_window, const void* k_new, const void* v_new,
                                      int window_len, int window_capacity, void* scores, int ld_scores, void* out, void* workspace,
                                      int Split_K, int T, int Batch_Size, int num_key_value_groups, float sqrt_d,
                                      const int32_t* window_len_extra, const void* attention_mask, int64_t mask_row_stride,
                                      int heads_per_mask_row, uint32_t flags, const int32_t* T_device);
int mustafar_decode_reads_extents(int num_key_value_groups, int ld_scores, uint32_t flags);

/* mustafar_decode_attention over two cache views (same semantics, same remaining arguments). */
int mustafar_decode_attention_view(void* stream, const mustafar_cache_view* k_cache, const mustafar_cache_view* v_cache,
                                   const void* q, void* k_window, void* v_window, const void* k_new, const void* v_new,
                                   int window_len, int window_capacity, void* scores, int ld_scores, void* out, void* workspace,
                                   int Split_K, int T, int Batch_Size, int num_key_value_groups, float sqrt_d,
                                   const int32_t* window_len_extra, const void* attention_mask, int64_t mask_row_stride,
                                   int heads_per_mask_row, uint32_t flags);

/*
 * In-place append of t new (already pruned) tokens per head behind the `old_tokens` a view holds: the cache-append
 * logic of the reference hook (models/llama_mustafar_kernel.py:339-390: shift the new offsets by the head's old total,
 * splice bitmaps / offsets, concatenate the streams) as two device passes that touch only the new tokens.
 *   pass 1  mustafar_cache_append_bitmap_{key,value}: bitmaps -> dst.bmp[h][2*old_tokens ...], offsets (continuing from
 *           dst.idx[h][2*old_tokens]; that entry must hold the head's current total, 0 for an empty cache) ->
 *           dst.idx[h][2*old_tokens + 1 ...]; head_total i64 [B'] = each head's NEW stream length in halfs.
 *           The caller checks head_total against the room behind nz_offset[h] before pass 2 (one small device->host read).
 *   pass 2  mustafar_cache_append_pack_{key,value}: the new tiles' non-zeros + padding -> dst.nz at 8*nz_offset[h] halfs.
 * x fp16 [B', t, D], t % 64 == 0, old_tokens % 64 == 0.
 */
/*
 * The two conversion calls of the reference (kernel/compression.py:249-432: bitmaps + offsets, then the packed streams) with ONE read of
 * the rows (round 5): the one-pass compression launch (kth = 0: already pruned) writes bitmaps, offsets and every head's stream -- into a
 * region of worst-case size each --, the caller reads head_off (the sizes of the tensors it returns: the one host read the reference's
 * return type asks for, compression.py:308, now BEHIND the work instead of between two passes over the rows) and
 * mustafar_convert_pack copies the regions head behind head into the exact-size buffer, the reference's layout.
 *   x          fp16 [B', t, 128] contiguous, t % 64 == 0           key        1: K tile geometry, 0: V
 *   bmp / accum   i64 [B', 2t] / i32 [B', 2t + 1] out (compression.py:339)
 *   head_off   i64 [B' + 1] out: exclusive prefix of the heads' stream lengths in halfs, [B'] = the total (compression.py:303-308)
 *   regions    B' * t * 128 halfs of scratch (head h's stream at h * t * 128);  packed  head_off[B'] halfs
 *   overflow_flag  device int, zeroed by the caller: bit 1 = a block gave up waiting for the lengths in front of it (see
 *              mustafar_cache_append_kv): the results are incomplete, repeat through mustafar_compress_bitmap_* / _pack_*
 *   scratch    mustafar_convert_scratch_bytes(B', t) bytes
 */
int64_t mustafar_convert_scratch_bytes(int Bp, int t);
int mustafar_convert_onepass(void* stream, const void* x, int Bp, int t, int D, int key, int64_t* bmp, int32_t* accum, int64_t* head_off,
                             void* regions, int32_t* overflow_flag, void* scratch);
/* The same with a host mirror as mustafar_compress_bitmap_mirrored takes one (round 6), B' + 2 int64 here: [0 .. B'] = head_off,
 * [B' + 1] = the value of *overflow_flag behind the compression launch (zero-extended).  Polled by the caller until no entry is negative. */
int mustafar_convert_onepass_mirrored(void* stream, const void* x, int Bp, int t, int D, int key, int64_t* bmp, int32_t* accum,
                                      int64_t* head_off, void* regions, int32_t* overflow_flag, void* scratch, int64_t* host_mirror);
int mustafar_convert_pack(void* stream, const void* regions, int Bp, int t, int D, const int64_t* head_off, void* packed);
int mustafar_cache_append_bitmap_key(void* stream, const void* x, int Bp, int t, int D, const mustafar_cache_view* dst,
                                     int old_tokens, int64_t* head_total);
int mustafar_cache_append_bitmap_value(void* stream, const void* x, int Bp, int t, int D, const mustafar_cache_view* dst,
                                       int old_tokens, int64_t* head_total);
int mustafar_cache_append_pack_key(void* stream, const void* x, int Bp, int t, int D, const mustafar_cache_view* dst,
                                   int old_tokens);
int mustafar_cache_append_pack_value(void* stream, const void* x, int Bp, int t, int D, const mustafar_cache_view* dst,
                                     int old_tokens);
int mustafar_counter_add(void* stream, int32_t* counter, int delta);

/*
 * The whole 256-token trigger of the hook (models/llama_mustafar_kernel.py:324-398) for K and V together, from the RAW
 * window rows: prune (:325-326, kth = max(1, int(sparsity * D)); 0 = rows already pruned), compress (:328-337) and append
 * (:339-390) in ONE launch behind a memset node: one wave per 64-token block keeps the rows in registers from the load to the
 * packed stream (threshold, bitmaps, counts, pack), and the blocks of a head find their stream positions from each other's
 * published lengths.  No pruned copy is written, nothing is allocated and nothing is read back on the host, so
 * the call can be captured in a hipGraph.  The same call compresses a prefill block (:416-437) into two empty views
 * (old_tokens = 0).  (MUSTAFAR_COMPRESS=twopass in the environment: the two-pass form, three launches.)
 *   k_x / v_x       fp16 rows of 128; head h starts `head_stride` elements after head h - 1 (a window buffer: capacity * 128)
 *   *_head_total    i64 [B'] out: every head's new stream length in halfs (read it whenever convenient, e.g. through an
 *                   asynchronous copy: the next append needs it only to decide about room)
 *   *_region_halfs  room of a head's stream region (0 = unchecked); overflow_flag (device int, REQUIRED; the caller zeroes it):
 *                   bit 0 = a head outgrew its region -- the blocks that would cross the end of the region then write no stream
 *                   bytes (the bitmaps / offsets of the new tokens and *_head_total, the length the head NEEDS, are written):
 *                   re-house at that length and repeat the call with the same old_tokens (it is idempotent);
 *                   bit 1 = a block gave up waiting for the lengths of the blocks in front of it and wrote nothing (the poll is
 *                   bounded so that every wave exits; it relies on workgroups being dispatched in order of their linear id, which
 *                   holds on gfx950; never seen): a hard error, the appended tokens are incomplete -- MUSTAFAR_COMPRESS=twopass in
 *                   the environment selects the three-launch form that has no such wait.
 *                   A caller that cannot read the flag right behind the launch (graph capture) keeps room for one worst-case
 *                   append (t * 128 halfs): bit 0 is then impossible.
 *   scratch         mustafar_compress_scratch_bytes(B', t) bytes
 */
int64_t mustafar_compress_scratch_bytes(int Bp, int t);
int mustafar_cache_append_kv(void* stream, const void* k_x, const void* v_x, int64_t head_stride, int Bp, int t, int D, int kth_k,
                             int kth_v, const mustafar_cache_view* k_dst, const mustafar_cache_view* v_dst, int old_tokens,
                             int64_t* k_head_total, int64_t* v_head_total, int64_t k_region_halfs, int64_t v_region_halfs,
                             int32_t* overflow_flag, void* scratch);
/*
 * Move a cache into another view (larger rows / regions: cache.py re-houses an arena when an append does not fit) -- the bitmaps
 * and offsets of the first `tokens` tokens and the first `stream_halfs` halfs of every head's stream, one launch.  `dst` has
 * equally spaced stream regions (nz_head_stride > 0); its nz_offset array is written.  tokens % 64 == 0, stream_halfs % 8 == 0.
 */
int mustafar_cache_rehouse(void* stream, const mustafar_cache_view* src, const mustafar_cache_view* dst, int Bp, int tokens,
                           int64_t stream_halfs);
/*
 * Form of the fused compression calls (mustafar_cache_append_kv, mustafar_trigger_compress_batch, mustafar_convert_onepass) from the next call
 * on: 0 = the process default (one pass unless MUSTAFAR_COMPRESS=twopass is in the environment), 1 = one pass, 2 = the two-pass form (three
 * launches, no wait between workgroups).  Round 5: a caller whose one-pass launch reported flag bit 1 (a block gave up waiting for the
 * lengths in front of it) repeats the call ONCE in the two-pass form -- the raw rows are still in place, the call is idempotent -- instead
 * of failing (cache.py; counted in cache.compress_fallbacks).  State of the calling host THREAD (round 6; it was process-wide): read the
 * current value (mustafar_compress_get_form), set it, call, set the old value back.
 * mustafar_compress_test_skip_publish (tests only; MUSTAFAR_EINVAL unless the process runs with MUSTAFAR_TEST_HOOKS=1): the block of this
 * index of the calling thread's NEXT one-pass launch does not publish its length, so that every block behind it times out (bit 1); -1 = off.
 */
int mustafar_compress_set_form(int form);
int mustafar_compress_get_form(void);
int mustafar_compress_test_skip_publish(int block);
/*
 * consolidate() on the device (round 5): the `n_extents` appended extents of a cache, read through its DEVICE table of views (the table
 * mustafar_decode_attention_extents reads), are copied behind the `base_tokens` tokens of `dst` -- a view with equally spaced stream regions
 * that mustafar_cache_rehouse has filled with the base in the launch in front of this one.  Offsets are shifted by the head's stream
 * length in front of each extent (model :352-360), read on the device; nothing is read on the host.  `max_extent_halfs`: an upper bound of
 * a head's stream length in any one extent (sizes the grid; the host knows every extent's lengths).  One launch; n_extents <= 21845.
 */
int mustafar_cache_consolidate_extents(void* stream, const mustafar_cache_view* dst, const mustafar_cache_view* extents, int n_extents, int Bp,
                                       int base_tokens, int64_t max_extent_halfs);
/* Window slide of the trigger (model :392-393) in place: rows [drop, len) of every head move to the front
 * (any number of rows may stay: overlapping ranges are moved in ascending pieces). */
int mustafar_window_drop_front(void* stream, void* k_window, void* v_window, int64_t head_stride, int Bp, int len, int drop);

/*
 * The 256-token trigger of ALL layers in two calls (round 4).  The reference runs the trigger layer by layer inside the attention
 * forward (models/llama_mustafar_kernel.py:324-398), each with a dozen host reads; here the layers' window rows are pruned and
 * compressed into one EXTENT each (mustafar_decode_attention_extents) by launches issued back to back from ONE call, the caller reads
 * every layer's flag and lengths with ONE copy, and a second call lists the extents in the device tables and slides the windows.
 * `items`: HOST array, one entry per layer.
 *   k_window / v_window   [B', capacity, 128] window buffers; rows [0, t) of every head are the raw tokens to compress
 *   k_dst / v_dst         views of the layer's new, EMPTY extent (nz_offset written, idx[h][0] == 0; nz_head_stride != 0)
 *   k_table_slot / v_table_slot   device addresses the views are written to by the finish call (the entry of the layer's extent
 *                         tables that the next decode launch will read), or NULL
 *   k_head_total / v_head_total / overflow_flag   as mustafar_cache_append_kv (the flags zeroed by the caller; one per layer)
 * mustafar_trigger_compress_batch: `scratch` = n x mustafar_compress_scratch_bytes(B', t) bytes; region_halfs as mustafar_cache_append_kv.
 * mustafar_trigger_finish_batch: `len` rows of every window are valid, `drop` (= t) leave; the rows that stay (any number) move to the front.
 */
typedef struct mustafar_trigger_item {
    void* k_window;
    void* v_window;
    mustafar_cache_view  k_dst, v_dst;
    mustafar_cache_view* k_table_slot;
    mustafar_cache_view* v_table_slot;
    int64_t* k_head_total;
    int64_t* v_head_total;
    int32_t* overflow_flag;
} mustafar_trigger_item;
int mustafar_trigger_compress_batch(void* stream, int n, const mustafar_trigger_item* items, int64_t head_stride, int Bp, int t, int D,
                                    int kth_k, int kth_v, int64_t k_region_halfs, int64_t v_region_halfs, void* scratch);
int mustafar_trigger_finish_batch(void* stream, int n, const mustafar_trigger_item* items, int64_t head_stride, int Bp, int len, int drop);

/*
 * FMA engine, process default (a fused call may carry its own in `flags`):
 *   2 = v_dot2_f32_f16 on pairs of tiles (default; GQA-4 one-pass decode launches).  One instruction does two tiles of one head, so
 *       the FMA phase costs 2 + 1.5 cheap instead of 4 vector instructions per tile.  For normal fp16 inputs the instruction is a
 *       two-term dot product with ONE rounding; a product with an fp16-SUBNORMAL input can lose up to its whole value (measured on
 *       gfx950, tools/ubench/dot2_asm_numerics.hip: under 1 % of such products do; never more than the product itself).  So the
 *       softmax weights travel scaled by 2^15 (weights down to 2^-29 stay normal; exactly undone when the slab is written), and a
 *       non-zero K / V / q element below 2^-14 contributes with an absolute error of at most 6.1e-5 x |coefficient| -- inside the
 *       fp16 rounding of the scores and outputs.  Kernels without a dot2 form (the two reference entry points, G < 4) run engine 0.
 *   0 = v_fma_mix_f32 per tile and head (exact fp16 x fp16 products in fp32, subnormals included)
 *   1 = v_mfma_f32_4x4x4_16B_f16 used as a 4-wide FMA unit (opt-in: the north_star leaves MFMA off)
 * Environment: MUSTAFAR_FMA_ENGINE=dot2|valu|mfma.  fp32 accumulation in every engine.
 */
int mustafar_set_fma_engine(int engine);
int mustafar_get_fma_engine(void);

/*
 * Structure of mustafar_decode_attention{,_view}: 1 = one-pass launch -- every wave runs key phase, softmax step and value
 * phase on its 64-token blocks and leaves (max, sum, unnormalised output) slabs that a row kernel merges (flash-decoding over
 * the compressed cache; needs ld_scores % 32 == 0, otherwise the other form runs); 0 = the round-1 form, key SpMV -> softmax
 * rows -> value SpMV -> sum; 2 (default) = the one-pass launch at every size and for every group count
 * (since round 3 it beats two launches at c2 .. c5 on all engines; the two-launch form runs when ld_scores % 32 != 0).  Also MUSTAFAR_ONEPASS=0|1|auto in the environment.  Same inputs, same
 * outputs within fp16 (the one-pass form normalises in fp32 at the end instead of rounding the probabilities to fp16).
 */
int mustafar_set_onepass(int mode);
int mustafar_get_onepass(void);
/* What the last fused call on the calling thread launched: FMA engine that ran (0 v_fma_mix_f32, 1 matrix pipe, 2 v_dot2_f32_f16)
 * | structure << 4 (0 two launches, 1 one-pass) | one-pass form << 8 (0 round-2 forms, 1 lean whole-block, 2 lean pair grain, 3 super-block pair form: round 5, the default from 768 workgroups on, 4 the small-launch form: round 6, the default below that);
 * -1 before the first call.  For tests and tools: a call's `flags` and the process defaults can be checked against what ran. */
int mustafar_last_decode_choice(void);
/* Tuning knobs for the measurement scripts under tools/ (launch shapes of the one-pass forms); not an operator interface.
 * Round 6: knob 11 = the small-launch kernel (0 never, 1 launches of two blocks per workgroup that put at most one wave on every SIMD -- the
 * default, 2 every launch of two blocks per workgroup); knob 12 = bytes of key stream per block for the speculative first-chunk request
 * (an experiment that measured slower; 0 = off, the default); knob 13 = Value_SplitK_API with N_Global = 8 on the lean kernel, the pad rows read by
 * workgroups of their own behind the row-0 workgroups (1, the default) or on round 1's kernel (0). */
int mustafar_tune(int knob, int value);

/*
 * Live kernel timing inside mustafar_decode_attention (bench.py roofline leg): HIP events that receive the start and
 * stop timestamps of the key and the value SpMV kernels themselves (hipExtLaunchKernel; the same interval rocprofv3
 * reports) for up to `max_records` calls.  mustafar_profile_end() waits for the recorded events, returns the average
 * durations in microseconds and releases the events.
 */
int mustafar_profile_begin(int max_records);
int mustafar_profile_end(double* key_us_avg, double* value_us_avg, int* records);
/* The same, and the average duration of the row kernel behind a one-pass launch (onepass_finish_kernel; 0 unless every record is one). */
int mustafar_profile_end2(double* key_us_avg, double* value_us_avg, double* finish_us_avg, int* records);

#ifdef __cplusplus
}
#endif
#endif /* MUSTAFAR_HIP_H */
