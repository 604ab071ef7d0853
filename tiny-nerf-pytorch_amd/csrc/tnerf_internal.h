// Internal definitions shared by the host planner and the HIP kernels of libtnerf_hip.so.
#pragma once
#include <stdint.h>
#include "../../include/tnerf.h"

#define TN_MAXD 16          // max hidden layers
#define TN_MAX_STEPS 32     // max input k-steps (pairs of input columns)
#define TN_JOB_INTS 16      // int32 per wgrad job record
#define TN_RED_HDR 260      // int32 header of the reduce table (1 + 64 classes * 4, padded)
#define TN_RED_MAXCLS 64

// Record layout of one wgrad job (one workgroup).
enum {
    JOB_A_ROW0 = 0, JOB_A_ROWS, JOB_B_ROW0, JOB_B_ROWS, JOB_N_AT, JOB_N_BT, JOB_WA,
    JOB_MBLK0, JOB_MBLKN, JOB_SLAB_OFF, JOB_CLASS, JOB_HAS_BIAS,
    JOB_A_BOUND, JOB_B_BOUND        // indices into the stash's bound words (TNB_*) of the rows' magnitude bounds
};
// Magnitude bounds of the stash's row groups, over all samples of the batch (fp32 bit patterns, atomicMax as unsigned; they sit
// behind the sign words of the stash: TN_BOUND_FLOATS words).  The x3 training forward / dgrad kernels maintain them (each sample's
// activation bound is what scales its fp16 pieces, mlpx3_core.hpp), the weight-gradient kernel scales the rows it splits into
// fp16 pieces by them.  The training forward's launcher clears them.
#define TN_BOUND_FLOATS 64
#define TNB_H(l) (l)                      // layer l's activations
#define TNB_ENC TN_MAXD                   // the network input
#define TNB_DZ(l) (TN_MAXD + 1 + (l))     // layer l's activation gradients
#define TNB_DZH (2 * TN_MAXD + 1)         // the head gradient
// The last bound word says WHICH matrix pipe's training forward filled the stash: the two pipes arrange the sign words (and, for the
// x3 pipe, the bound words themselves) differently, so a backward kernel of the other pipe must not consume it.  Written by the
// training forward (x3: by the kernel; fp32 MFMA: by its launcher), checked by every dgrad and weight-gradient kernel: on a mismatch
// the kernel marks the tag BAD and the weight-gradient kernel fills its slabs with NaN — a loud failure instead of silently wrong
// gradients.  tnerf_wgrad (the per-call entry point) picks its kernel body by the tag.
#define TNB_TAG (TN_BOUND_FLOATS - 1)
#define TN_TAG_X3  0x78330004u
#define TN_TAG_F32 0x66333204u
#define TN_TAG_BAD 0x0bad0badu

// Everything the kernels need to know about one model; passed by value as a kernel argument.
struct MlpLayout {
    int32_t in_dim, hidden, depth, skip_at;
    int32_t flags;              // TNERF_FLAG_*
    int32_t NE;                 // input k-steps (20 or 32)
    int32_t NT;                 // hidden/32
    int16_t emap[TN_MAX_STEPS][2];
    // packed-weight segment offsets (floats); -1 = absent
    int64_t fw_bias[TN_MAXD], fw_enc[TN_MAXD], fw_hid[TN_MAXD];
    int64_t fw_head_bias, fw_head;
    int64_t bw_hid[TN_MAXD], bw_head;
    int64_t packed_floats;
    // flat parameter offsets
    int64_t p_w[TN_MAXD], p_b[TN_MAXD];
    int32_t fan_in[TN_MAXD];
    int64_t p_ws, p_bs, p_wc, p_bc, n_params;        // n_params: of the caller's (true-width) model; the offsets above: PADDED space
    int32_t hidden_true;        // the caller's hidden width (<= hidden, the kernel width 128 / 256 it is zero-padded to)
    int64_t n_params_padded;
    // stash rows (feature-major matrices, row stride Mp floats)
    int32_t enc_row0, h_row0[TN_MAXD], out_row0, dz_row0[TN_MAXD], dzh_row0, stash_rows;
};

// bf16 mode (mlp16_*.hip): the weights are one STREAM of 1 KB MFMA A-fragments (v_mfma_f32_32x32x16_bf16: lane l holds
// 8 bf16 = W[row l&31][k-slot 8(l>>5)+e]) in exactly the order a wavefront consumes them during one pass over the
// network, cut into stages of TN16_STAGE fragments; a workgroup streams the stages through an LDS ring.  Order:
//   layer 0      : for n-tile t: TN16_KE input k-steps
//   hidden layer : for n-tile t: hidden/16 k-steps           (skip layer: then TN16_KE input k-steps)
//   heads        : hidden/16 k-steps of the one head tile (rows r,g,b,sigma), zero-padded to a whole stage
// followed by the fp32 biases (depth*hidden hidden-layer biases in natural order, then r,g,b,sigma).
// K-slot <-> feature maps (what makes an accumulator tile the next layer's B operand with no data movement):
//   hidden k-step s, lane-half h, element e : feature 32(s>>1) + TN_ACC_ROW(8(s&1)+e, h)
//   input  k-step u, lane-half h, element e : slot a = 8u+e;  a < 3L: (sin, cos)[h](2^(a/3) x_(a%3));  a = 3L: (x, y)[h];
//                                             a = 3L+1: (z, 0)[h];  else 0          (L = (in_dim-3)/6 <= 10)
#define TN16_STAGE 16
#define TN16_KE 4
// Backward stream (dgrad), after the forward one: W_head^T (one fragment per k-tile: row 32t+i, k-slot (h=0, e<4) <->
// head row e = r,g,b,sigma; zero-padded to a stage), then for l = depth-1 .. 1 the hidden part of W_l transposed: for
// k-tile t: hidden/16 k-steps, fragment row 32t+i <-> input feature, k-slot (s,h,e) <-> output feature as above.
//
// bf16 training stash, organised by TILES of 32 sample slots (tile = ray * ceil(S/32) + sb/32; slots past S are zero):
//   fragments: tile * n_ft + ft  ->  2 KB = [k-step u: 2][lane 64][8 bf16], the K = SAMPLES operand of the weight-gradient
//              MFMAs: lane (c, h) element e = value of feature (32 ft' + c) for sample slot TN_ACC_ROW(8u+e, h).
//              ft: ft_enc + {0,1} (input slots: tile T = u>>1, c = TN_ACC_ROW(8(u&1)+e, h) of slot (u,h,e)),
//                  ft_h[l] + t (H_l), ft_dz[l] + t (dZ_l), ft_dzh (rows 0..3 = d r,g,b,sigma pre-activation)
//   masks    : [layer][tile][lane 64][hidden/64 words]: bit (t&1)*16 + r of word t/2 <-> H_l[feature 32t + TN_ACC_ROW(r,h)] > 0
//   out4     : [tile][slot 32][4] fp32 head outputs (r,g,b after sigmoid, sigma after ReLU)
// One extra ("dump") tile follows the real ones in every region: waves that pad the last workgroup store there.
struct Net16 {
    int32_t in_dim, hidden, depth, skip_at, Lf;
    int32_t n_frag, n_stage;      // forward stream, per pass
    int32_t n_bw_frag, n_bw_stage;// backward stream, per pass (starts at byte n_frag * 1024)
    int32_t bias_off;             // byte offset of the fp32 biases inside the packed buffer
    int32_t n_bias;               // depth*hidden + 4
    int32_t ft_enc, ft_h[TN_MAXD], ft_dz[TN_MAXD], ft_dzh, n_ft;
    int64_t packed_bytes;
    int64_t pack_entries;         // (n_frag + n_bw_frag)*512 + n_bias
};
#define TN16_FT_BYTES 2048
#define TN16_STASH_FRAG_BYTES(n, tiles) ((int64_t)((tiles) + 1) * (n).n_ft * TN16_FT_BYTES)
#define TN16_STASH_MASK_BYTES(n, tiles) ((int64_t)(n).depth * ((tiles) + 1) * 64 * ((n).hidden / 64) * 4)
#define TN16_STASH_OUT_BYTES(n, tiles)  ((int64_t)((tiles) + 1) * 32 * 16)

// "x3" chain kernels (mlpx3.hip): the fp32 MLP chain on the fp16 matrix pipe ("three partial products").  Every
// fp32 operand is carried as TWO fp16 pieces of its power-of-two-scaled value (round to nearest: x 2^s = p1 + p2 up to 2^-22
// relative), a product is the three partial products p1 q1 (-> main accumulator), p1 q2 + p2 q1 (-> correction accumulator),
// both accumulators fp32.  The weights are a STREAM of k-step RECORDS in the order a wavefront consumes them.  A layer is walked
// as two HALF-PASSES (output tiles 0..NT/2-1, then NT/2..NT-1: the epilogue of one half hides behind the MFMAs of the other);
// a record holds, for every tile slot tl of ONE half, the two fp16 pieces of the A fragment of (n-tile half*NT/2 + tl, this k-step):
//   record = [tl = 0 .. NT/2-1][piece 0..1] x 1 KB,   fragment = lane (row l&31, half l>>5) x 8 fp16 (k-slots 8 (l>>5) + e)
// Forward order: layer 0: half A: TN16_KE input k-steps, half B: the same; layer l >= 1: per half hidden/16 hidden k-steps
// (+ TN16_KE input k-steps for the skip layer); heads: ONE output tile (rows r,g,b,sigma), so the NT/2 slots of a record carry
// NT/2 consecutive k-steps of it: hidden/16 / (NT/2) records.  k-slot <-> feature maps as in the bf16 mode (above).  Records are cut into stages of
// TX_STAGE fragments (16 KB: two records of a 256-wide net, four of a 128-wide one) for the LDS ring.  After the stream: the
// fp32 biases as in the bf16 mode, then TX_META floats per layer (index depth = the heads):
//   [0] 2^-s   : what the layer's pieces in the stream must be multiplied by to give the weights (the scale IN EFFECT)
//   [1] max|W| , [2] max|b| of the layer as of the last k_x3stats (the chain kernels bound their activations with them)
//   [3] 2^s'   : the scale the NEXT (re)pack of the layer uses, chosen from [1] so that max|W| 2^s' is in [2^11, 2^12)
//   [4], [5]   : running maxima (fp32 bit patterns, atomicMax as unsigned) of |W|, |b| that whoever rewrites the layer's
//                parameters accumulates (the finishing kernel; k_x3stats_scan before a full pack); k_x3stats_final turns them
//                into [1], [2], [3] and clears them
#define TX_STAGE 16
#define TX_NP 2
#define TX_META 8
struct NetX3 {
    int32_t in_dim, hidden, depth, skip_at, Lf;
    int32_t NT, KH;               // n-tiles (hidden/32), hidden k-steps (hidden/16)
    int32_t rec_frags;            // NT / 2 * TX_NP
    int32_t n_rec, n_stage;       // forward stream, per pass
    int32_t n_bw_rec, n_bw_stage; // backward (dgrad) stream, per pass: starts at byte n_rec * rec_frags * 1024
    int32_t bias_off, n_bias;     // byte offset / count of the fp32 biases (depth*hidden + 4)
    int32_t meta_off;             // byte offset of the (depth + 1) * TX_META scale floats (= bias_off + 4 n_bias: contiguous with the biases)
    int32_t fw_rec0[TN_MAXD + 2]; // first forward record of layer l; [depth] = heads; [depth + 1] = n_rec
    int64_t packed_bytes, pack_entries;     // pack_entries = (n_rec + n_bw_rec) * rec_frags * 512 + n_bias
};
// Backward stream: heads^T (two records, half A and half B: fragment row 32t+i <-> feature, k-slot (h=0, e<4) <-> head row
// e = r,g,b,sigma; padded to a whole stage), then for l = depth-1 .. 1 the hidden part of W_l transposed: per half hidden/16
// records, fragment row 32t+i <-> input feature, k-slot (s,h,e) <-> output feature.
// Layer of a stream record (forward or backward numbering continued behind n_rec); depth = heads.
#ifdef __HIPCC__
#define TN_HD __host__ __device__
#else
#define TN_HD
#endif
TN_HD static inline int tx_record_layer(const NetX3* n, int rec) {
    if (rec < n->n_rec) { int l = 0; while (rec >= n->fw_rec0[l + 1]) ++l; return l; }
    const int r = rec - n->n_rec, rps = TX_STAGE / n->rec_frags;
    return r < rps ? n->depth : n->depth - 1 - (r - rps) / (2 * n->KH);
}

#ifdef __cplusplus
extern "C" {
#endif
int  tn_build_netx3(const tnerf_mlp_desc* d, NetX3* n);        // 0 or TNERF_E*
// host_plan.cpp
int  tn_build_layout(const tnerf_mlp_desc* d, MlpLayout* L);   // 0 or TNERF_E*
int  tn_build_net16(const tnerf_mlp_desc* d, Net16* n);        // 0 or TNERF_E*
void tn_set_error(const char* fmt, ...);
int  tn_check_ray_ws(const char* who, int64_t n_rays, int64_t ws_floats);                        // 0 or TNERF_ESMALL
int  tn_check_stash32(const char* who, const tnerf_mlp_desc* d, int64_t n_samples_total, int64_t stride, int64_t capacity_floats);
int  tn_check_stash16(const char* who, const tnerf_mlp_desc* d, int64_t n_rays, int32_t n_samples, int64_t capacity_bytes);
#ifdef __cplusplus
}
#endif

// The training stash is BLOCK-major: samples are grouped in blocks of 32 consecutive indices and block b holds
//   stash[(b * stash_rows + row) * 32 + (m & 31)],   b = m >> 5
// i.e. one 128-byte line per (block, row): a wave's tile of one layer is 32 KB of contiguous HBM for both the
// writer (forward / dgrad kernels) and the reader (wgrad), instead of 256 lines that are 1 MB apart.
// After the Mp/32 blocks come the ReLU sign bits of
// every hidden layer: for layer l, sample m, lane-half h: hidden/64 uint32 words
//   word index ((l*(Mp+32) + m)*2 + h) * (hidden/64) + t/2,  bit (t&1)*16 + r   <->  feature 32t + TN_ACC_ROW(r,h)
// One extra ("dump") block / sample slot follows the Mp real ones: lanes that pad a ragged tile store there, so
// that no store in the hot loops needs a per-lane branch.  (The x3 chain kernels' H / dZ rows do not use it: their stores go
// through a per-tile buffer resource whose range check drops padding lanes — mlpx3_core.hpp TxDst; sign words, input rows and
// head rows still do.)
#define TN_RAY_WS_FLOATS 4                               // floats per ray of the train steps' workspace (tnerf_train_ws_floats)
#define TN_STASH_BODY_FLOATS(L, Mp) ((int64_t)(L).stash_rows * ((Mp) + 32))
#define TN_MASK_FLOATS(L, Mp) ((int64_t)(L).depth * ((Mp) + 32) * ((L).hidden / 32))
#define TN_BOUND_OFF(L, Mp) (TN_STASH_BODY_FLOATS(L, Mp) + TN_MASK_FLOATS(L, Mp))      // float offset of the bound words

// row of a 32x32 MFMA accumulator register r (0..15) for lane-half h (0/1)
#define TN_ACC_ROW(r, h) (((r) & 3) + 8 * ((r) >> 2) + 4 * (h))
