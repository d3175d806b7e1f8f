// Fragment packs of one DCGRU cell: layout (CellPack) and the element map (pack_cell_body), shared by the single-cell launch
// (kernels_pack.h pack_cell_kernel) and the all-cells launch (spec_inst.cpp pack_cells_kernel).  Layout conventions: kernels_pack.h.
#pragma once
#include "common.h"
#include "nnq_order.h"

namespace eeg {

// Offsets (in floats) of the packs of one DCGRU cell inside a single device buffer.
struct CellPack {
    int Fin, H, M;
    size_t bx;     // x-part, K = M*Fin,  O = 3H   [gate(2H) | cand(H)]          (fwd hoisted GEMM)
    size_t bias;   // 3H                            [bg | bc]
    size_t bhg;    // h-part gate, K = M*H,  O = 2H                               (fwd recurrence)
    size_t bhc;    // h-part cand, K = M*H,  O = H                                (fwd recurrence)
    size_t b1;     // bwd cand:  K = M*H  (k = m*H + o),  O = H (f)   = Wc^h transposed
    size_t b2;     // bwd gate:  K = M*2H (k = m*2H + o), O = H (f)   = Wg^h transposed
    size_t bxt;    // bwd dx:    K = 3H (k = o), O = round_up(M*Fin,16)  = Bx transposed
    // persistent decoder backward (kernels_decoder.h): b1 / b2 widened by the input-feature columns, so that the
    // recurrent GEMMs also produce dX = sum_m (P_m^T dXW) W^x_m^T from the SAME adjoint hop rows
    // (O = cell_pack_cx_cols: 12 column tiles for 64 units and up to 128 input features -- ONE literal tile count for every
    //  layer of the decoder, so the streamed-weight addresses are base + immediate)
    size_t c1;     // K = M*H  (k = m*H + o),  O columns: [Wc^h | Wc^x | 0] transposed, quad-permuted K
    size_t c2;     // K = M*2H (k = m*2H + o), O columns: [Wg^h | Wg^x | 0] transposed, quad-permuted K
    // round 3, gemm_nnr_kernel (kernels_gemm_q.h): the same two right-hand sides in quad order (one ds_read_b128 per lane feeds
    // the four MFMAs of a 16-deep K chunk; chunk order of make_nnq_order); bxtq exists when M*Fin is a multiple of 192
    size_t bxq;    // x-part:  nnq order over (M planes x Fin), 3H/16 column tiles
    size_t bxtq;   // bwd dx:  nnq order over (1 segment x 3H), M*Fin/16 column tiles (0 floats when not applicable)
    bool has_bxq, has_bxtq;
    size_t total;
};

__host__ __device__ inline CellPack make_cell_pack(int Fin, int H, int M) {
    CellPack p;
    p.Fin = Fin; p.H = H; p.M = M;
    size_t o = 0;
    p.bx = o;   o += (size_t)M * Fin * 3 * H;
    p.bias = o; o += (size_t)round_up(3 * H, 64);
    p.bhg = o;  o += (size_t)M * H * 2 * H;
    p.bhc = o;  o += (size_t)M * H * H;
    p.b1 = o;   o += (size_t)M * H * H;
    p.b2 = o;   o += (size_t)M * 2 * H * H;
    p.bxt = o;  o += (size_t)3 * H * round_up(M * Fin, 16);
    p.c1 = o;   o += (size_t)M * H * cell_pack_cx_cols(Fin, H);
    p.c2 = o;   o += (size_t)M * 2 * H * cell_pack_cx_cols(Fin, H);
    p.has_bxq = (3 * H) % 192 == 0;     // (any tail count: gemm_nnr_kernel takes <= 2 tail chunks, the decoder kernels any)
    p.has_bxtq = (M * Fin) % 192 == 0 && (3 * H) % 4 == 0 && make_nnq_order(1, 3 * H).ntail <= 2;
    p.bxq = o;  o += p.has_bxq ? (size_t)round_up(make_nnq_order(M, Fin).nch, 4) * (3 * H / 16) * 256 : 0;   // (zero chunks up to a multiple of 4: kernels_decoder.h gemm_stream_nnq)
    p.bxtq = o; o += p.has_bxtq ? (size_t)make_nnq_order(1, 3 * H).nch * (M * Fin / 16) * 256 : 0;
    p.total = o;
    return p;
}

// element (k, j) of each logical B matrix, read from the reference-layout tensors
__device__ __forceinline__ float ref_wg(const float* Wg, int M, int H, int f_all, int m, int o) {
    return Wg[((size_t)f_all * M + m) * (2 * H) + o];
}
__device__ __forceinline__ float ref_wc(const float* Wc, int M, int H, int f_all, int m, int o) {
    return Wc[((size_t)f_all * M + m) * H + o];
}

// the packs of one cell, by the workgroups bid of nb (grid-stride over the elements of the block)
__device__ __forceinline__ void pack_cell_body(const float* __restrict__ Wg, const float* __restrict__ bg,
                                               const float* __restrict__ Wc, const float* __restrict__ bc,
                                               float* __restrict__ out, const CellPack& p, int bid, int nb) {
    const int Fin = p.Fin, H = p.H, M = p.M;
    const size_t stride = (size_t)nb * blockDim.x;
    for (size_t idx = (size_t)bid * blockDim.x + threadIdx.x; idx < p.total; idx += stride) {
        float v = 0.f;
        if (idx < p.bias) {                       // bx: NCT = 3H/16
            const size_t e = idx - p.bx;
            const int lane = e & 63, nct = 3 * H / 16;
            const int ct = (e >> 6) % nct, ks = (e >> 6) / nct;
            const int k = 4 * ks + (lane >> 4), j = 16 * ct + (lane & 15);
            const int m = k / Fin, f = k % Fin;
            v = j < 2 * H ? ref_wg(Wg, M, H, f, m, j) : ref_wc(Wc, M, H, f, m, j - 2 * H);
        } else if (idx < p.bhg) {                 // bias
            const int j = idx - p.bias;
            v = j < 2 * H ? bg[j] : (j < 3 * H ? bc[j - 2 * H] : 0.f);
        } else if (idx < p.bhc) {                 // bhg: NCT = 2H/16
            const size_t e = idx - p.bhg;
            const int lane = e & 63, nct = 2 * H / 16;
            const int ct = (e >> 6) % nct, ks = (e >> 6) / nct;
            const int k = kperm(ks, lane >> 4), j = 16 * ct + (lane & 15);
            v = ref_wg(Wg, M, H, Fin + k % H, k / H, j);
        } else if (idx < p.b1) {                  // bhc: NCT = H/16
            const size_t e = idx - p.bhc;
            const int lane = e & 63, nct = H / 16;
            const int ct = (e >> 6) % nct, ks = (e >> 6) / nct;
            const int k = kperm(ks, lane >> 4), j = 16 * ct + (lane & 15);
            v = ref_wc(Wc, M, H, Fin + k % H, k / H, j);
        } else if (idx < p.b2) {                  // b1[k = m*H + o][f]
            const size_t e = idx - p.b1;
            const int lane = e & 63, nct = H / 16;
            const int ct = (e >> 6) % nct, ks = (e >> 6) / nct;
            const int k = kperm(ks, lane >> 4), f = 16 * ct + (lane & 15);
            v = ref_wc(Wc, M, H, Fin + f, k / H, k % H);
        } else if (idx < p.bxt) {                 // b2[k = m*2H + o][f]
            const size_t e = idx - p.b2;
            const int lane = e & 63, nct = H / 16;
            const int ct = (e >> 6) % nct, ks = (e >> 6) / nct;
            const int k = kperm(ks, lane >> 4), f = 16 * ct + (lane & 15);
            v = ref_wg(Wg, M, H, Fin + f, k / (2 * H), k % (2 * H));
        } else if (idx < p.c1) {                  // bxt[k = o][j = m*Fin + f]
            const size_t e = idx - p.bxt;
            const int lane = e & 63, nct = round_up(M * Fin, 16) / 16;
            const int ct = (e >> 6) % nct, ks = (e >> 6) / nct;
            const int o = 4 * ks + (lane >> 4), j = 16 * ct + (lane & 15);
            if (j < M * Fin) {
                const int m = j / Fin, f = j % Fin;
                v = o < 2 * H ? ref_wg(Wg, M, H, f, m, o) : ref_wc(Wc, M, H, f, m, o - 2 * H);
            }
        } else if (idx >= p.bxq) {                // quad packs of gemm_nnr_kernel: [(c * nct + ct) * 64 + lane][s]
            const bool tr = idx >= p.bxtq;
            const size_t e = idx - (tr ? p.bxtq : p.bxq);
            const int s4 = e & 3, lane = (e >> 2) & 63, nct = tr ? M * Fin / 16 : 3 * H / 16;
            const int ct = (e >> 8) % nct, c = (e >> 8) / nct, j = 16 * ct + (lane & 15);
            const int k = nnq_k_of(tr ? make_nnq_order(1, 3 * H) : make_nnq_order(M, Fin), c, lane >> 4, s4);
            if (k >= 0) {
                if (!tr) {                        // W^x[k = m*Fin + f][j]
                    const int m = k / Fin, f = k % Fin;
                    v = j < 2 * H ? ref_wg(Wg, M, H, f, m, j) : ref_wc(Wc, M, H, f, m, j - 2 * H);
                } else {                          // (W^x)^T[k = o][j = m*Fin + f]
                    const int m = j / Fin, f = j % Fin;
                    v = k < 2 * H ? ref_wg(Wg, M, H, f, m, k) : ref_wc(Wc, M, H, f, m, k - 2 * H);
                }
            }
        } else {                                  // c1 / c2 [k = m*W + o][j]: j < H hidden feature j, else input feature j - H
            const bool gate = idx >= p.c2;
            const size_t e = idx - (gate ? p.c2 : p.c1);
            const int lane = e & 63, nct = cell_pack_cx_cols(Fin, H) / 16, W = gate ? 2 * H : H;
            const int ct = (e >> 6) % nct, ks = (e >> 6) / nct;
            const int k = kperm(ks, lane >> 4), j = 16 * ct + (lane & 15);
            const int f_all = j < H ? Fin + j : j - H;
            if (j < H + Fin) v = gate ? ref_wg(Wg, M, H, f_all, k / W, k % W) : ref_wc(Wc, M, H, f_all, k / W, k % W);
        }
        out[idx] = v;
    }
}
}  // namespace eeg
