// Micro-benchmark: what does one step of the position-parallel header walk (decode_seg.hip, seg_walk) cost, and why?
// One wavefront per SIMD-slot runs N steps of the hand-scheduled counting loop over a private LDS row of random bits.
//   hipcc --offload-arch=gfx950 -O3 -o tools/stepbench tools/stepbench.hip && tools/stepbench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

constexpr int kRow = 36;

template <int VARIANT>
__global__ __launch_bounds__(256) void k_step(const uint32_t* __restrict__ seed, uint32_t steps, uint64_t* __restrict__ out) {
    __shared__ uint32_t win[4][64 * kRow];
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    for (int i = 0; i < kRow; ++i) win[wave][lane * kRow + i] = seed[(blockIdx.x * 256 + threadIdx.x) * kRow + i];
    __syncthreads();
    uint32_t pos = lane * 3u, w = 2u, n = 0u;
    const uint32_t rowb = (uint32_t)(uintptr_t)(&win[wave][lane * kRow]);
    const uint32_t endx = 0xF0000000u, wend = 0xF0000000u, w0c = 0u, maxw = 16u;
    uint64_t dmask = 0ull, zr = 0ull, t_act, t_sa, t_ex;
    uint32_t t_li, t_a, t_bits, t_w3, t_wa, t_wb, t_hx, t_t, cnt = steps;
    const uint64_t r0 = wall_clock64();
    const uint64_t t0 = clock64();
    if (VARIANT == 0) {
        asm volatile(
            "s_mov_b64 %[ex], exec\n"
            "1:\n\t"
            "v_cmp_gt_u32 vcc, %[wend], %[pos]\n\t"
            "s_andn2_b64 %[act], vcc, %[done]\n\t"
            "s_cbranch_scc0 9f\n\t"
            "s_mov_b64 exec, %[act]\n\t"
            "v_subrev_u32 %[li], %[w0c], %[pos]\n\t"
            "v_and_b32 %[li], 0x3ff, %[li]\n\t"
            "v_lshrrev_b32 %[a], 5, %[li]\n\t"
            "v_lshl_add_u32 %[a], %[a], 2, %[rowb]\n\t"
            "ds_read2_b32 v[62:63], %[a] offset1:1\n\t"
            "s_waitcnt lgkmcnt(0)\n\t"
            "v_alignbit_b32 %[bits], v63, v62, %[li]\n\t"
            "v_bfe_u32 %[w3], %[bits], 1, 3\n\t"
            "v_bfe_u32 %[wa], %[bits], 4, 2\n\t"
            "v_bfe_u32 %[wb], %[bits], 6, 6\n\t"
            "v_cmp_eq_u32 %[sa], 7, %[w3]\n\t"
            "v_cmp_eq_u32 vcc, 3, %[wa]\n\t"
            "v_add_u32 %[wa], 7, %[wa]\n\t"
            "v_add_u32 %[wb], 10, %[wb]\n\t"
            "v_cndmask_b32 %[wa], %[wa], %[wb], vcc\n\t"
            "v_cndmask_b32 %[hx], 6, 12, vcc\n\t"
            "v_cndmask_b32 %[wa], %[w3], %[wa], %[sa]\n\t"
            "v_cndmask_b32 %[hx], 4, %[hx], %[sa]\n\t"
            "v_and_b32 %[t], 1, %[bits]\n\t"
            "v_cmp_eq_u32 vcc, 1, %[t]\n\t"
            "v_cndmask_b32 %[w], %[wa], %[w], vcc\n\t"
            "v_cndmask_b32 %[hx], %[hx], 1, vcc\n\t"
            "v_cmp_lt_u32 %[sa], %[maxw], %[w]\n\t"
            "v_cndmask_b32 %[w], %[w], 0, %[sa]\n\t"
            "v_cmp_eq_u32 %[sa], 0, %[w]\n\t"
            "s_and_b64 %[zr], %[sa], vcc\n\t"
            "s_nop 0\n\t"                                   // (the real loop branches to the zero-run path here)
            "v_mad_u32_u24 %[t], %[w], 12, %[hx]\n\t"
            "v_add_u32 %[pos], %[pos], %[t]\n\t"
            "v_add_u32 %[n], 1, %[n]\n\t"
            "v_cmp_le_u32 vcc, %[endx], %[pos]\n\t"
            "s_or_b64 %[done], %[done], vcc\n\t"
            "s_mov_b64 exec, %[ex]\n\t"
            "s_sub_u32 %[cnt], %[cnt], 1\n\t"
            "s_cbranch_scc0 1b\n"                           // (cnt wraps below zero: scc = borrow)
            "9:\n"
            : [pos] "+v"(pos), [w] "+v"(w), [n] "+v"(n), [done] "+s"(dmask), [zr] "+s"(zr), [act] "=&s"(t_act), [sa] "=&s"(t_sa),
              [ex] "=&s"(t_ex), [li] "=&v"(t_li), [a] "=&v"(t_a), [bits] "=&v"(t_bits), [w3] "=&v"(t_w3), [wa] "=&v"(t_wa),
              [wb] "=&v"(t_wb), [hx] "=&v"(t_hx), [t] "=&v"(t_t), [cnt] "+s"(cnt)
            : [wend] "s"(wend), [w0c] "s"(w0c), [rowb] "v"(rowb), [maxw] "s"(maxw), [endx] "v"(endx)
            : "vcc", "scc", "memory", "v62", "v63");
    } else if (VARIANT == 1) {                              // no LDS read: the bits come from a register
        asm volatile(
            "s_mov_b64 %[ex], exec\n"
            "1:\n\t"
            "v_cmp_gt_u32 vcc, %[wend], %[pos]\n\t"
            "s_andn2_b64 %[act], vcc, %[done]\n\t"
            "s_cbranch_scc0 9f\n\t"
            "s_mov_b64 exec, %[act]\n\t"
            "v_subrev_u32 %[li], %[w0c], %[pos]\n\t"
            "v_and_b32 %[li], 0x3ff, %[li]\n\t"
            "v_lshrrev_b32 %[a], 5, %[li]\n\t"
            "v_lshl_add_u32 %[a], %[a], 2, %[rowb]\n\t"
            "v_mul_u32_u24 %[bits], 0x9e3779, %[a]\n\t"
            "v_alignbit_b32 %[bits], %[bits], %[a], %[li]\n\t"
            "v_bfe_u32 %[w3], %[bits], 1, 3\n\t"
            "v_bfe_u32 %[wa], %[bits], 4, 2\n\t"
            "v_bfe_u32 %[wb], %[bits], 6, 6\n\t"
            "v_cmp_eq_u32 %[sa], 7, %[w3]\n\t"
            "v_cmp_eq_u32 vcc, 3, %[wa]\n\t"
            "v_add_u32 %[wa], 7, %[wa]\n\t"
            "v_add_u32 %[wb], 10, %[wb]\n\t"
            "v_cndmask_b32 %[wa], %[wa], %[wb], vcc\n\t"
            "v_cndmask_b32 %[hx], 6, 12, vcc\n\t"
            "v_cndmask_b32 %[wa], %[w3], %[wa], %[sa]\n\t"
            "v_cndmask_b32 %[hx], 4, %[hx], %[sa]\n\t"
            "v_and_b32 %[t], 1, %[bits]\n\t"
            "v_cmp_eq_u32 vcc, 1, %[t]\n\t"
            "v_cndmask_b32 %[w], %[wa], %[w], vcc\n\t"
            "v_cndmask_b32 %[hx], %[hx], 1, vcc\n\t"
            "v_cmp_lt_u32 %[sa], %[maxw], %[w]\n\t"
            "v_cndmask_b32 %[w], %[w], 0, %[sa]\n\t"
            "v_cmp_eq_u32 %[sa], 0, %[w]\n\t"
            "s_and_b64 %[zr], %[sa], vcc\n\t"
            "s_nop 0\n\t"
            "v_mad_u32_u24 %[t], %[w], 12, %[hx]\n\t"
            "v_add_u32 %[pos], %[pos], %[t]\n\t"
            "v_add_u32 %[n], 1, %[n]\n\t"
            "v_cmp_le_u32 vcc, %[endx], %[pos]\n\t"
            "s_or_b64 %[done], %[done], vcc\n\t"
            "s_mov_b64 exec, %[ex]\n\t"
            "s_sub_u32 %[cnt], %[cnt], 1\n\t"
            "s_cbranch_scc0 1b\n"
            "9:\n"
            : [pos] "+v"(pos), [w] "+v"(w), [n] "+v"(n), [done] "+s"(dmask), [zr] "+s"(zr), [act] "=&s"(t_act), [sa] "=&s"(t_sa),
              [ex] "=&s"(t_ex), [li] "=&v"(t_li), [a] "=&v"(t_a), [bits] "=&v"(t_bits), [w3] "=&v"(t_w3), [wa] "=&v"(t_wa),
              [wb] "=&v"(t_wb), [hx] "=&v"(t_hx), [t] "=&v"(t_t), [cnt] "+s"(cnt)
            : [wend] "s"(wend), [w0c] "s"(w0c), [rowb] "v"(rowb), [maxw] "s"(maxw), [endx] "v"(endx)
            : "vcc", "scc", "memory", "v62", "v63");
    } else if (VARIANT == 3) {                              // software-pipelined: the next read is issued as soon as the position is known
        uint32_t pw = 8u * rowb + pos, stop = 0xF0000000u, ls = 1u + 12u * w;
        const uint32_t c90 = 90u, c132 = 132u;
        asm volatile(
            "s_mov_b64 %[ex], exec\n\t"
            "v_cmp_lt_u32 vcc, %[pw], %[stop]\n\t"
            "s_and_b64 exec, exec, vcc\n\t"
            "s_cbranch_scc0 9f\n\t"
            "v_lshrrev_b32 %[a], 3, %[pw]\n\t"
            "v_and_b32 %[a], 0xffc, %[a]\n\t"
            "ds_read2_b32 v[62:63], %[a] offset1:1\n"
            "1:\n\t"
            "s_waitcnt lgkmcnt(0)\n\t"
            "v_alignbit_b32 %[bits], v63, v62, %[pw]\n\t"
            "v_bfe_u32 %[w3], %[bits], 1, 3\n\t"
            "v_bfe_u32 %[wa], %[bits], 4, 2\n\t"
            "v_bfe_u32 %[wb], %[bits], 6, 6\n\t"
            "v_and_b32 %[t], 1, %[bits]\n\t"
            "v_mad_u32_u24 %[lx], %[w3], 12, 4\n\t"
            "v_mad_u32_u24 %[t1], %[wa], 12, %[c90]\n\t"
            "v_mad_u32_u24 %[t2], %[wb], 12, %[c132]\n\t"
            "v_cmp_eq_u32 vcc, 3, %[wa]\n\t"
            "v_cndmask_b32 %[t1], %[t1], %[t2], vcc\n\t"
            "v_add_u32 %[wa], 7, %[wa]\n\t"
            "v_add_u32 %[wb], 10, %[wb]\n\t"
            "v_cndmask_b32 %[wa], %[wa], %[wb], vcc\n\t"
            "v_cmp_eq_u32 vcc, 7, %[w3]\n\t"
            "v_cndmask_b32 %[lx], %[lx], %[t1], vcc\n\t"
            "v_cndmask_b32 %[w3], %[w3], %[wa], vcc\n\t"
            "v_cmp_eq_u32 vcc, 1, %[t]\n\t"
            "v_cndmask_b32 %[lx], %[lx], %[ls], vcc\n\t"
            "v_add_u32 %[pw], %[pw], %[lx]\n\t"
            "v_lshrrev_b32 %[a], 3, %[pw]\n\t"
            "v_and_b32 %[a], 0xffc, %[a]\n\t"
            "ds_read2_b32 v[62:63], %[a] offset1:1\n\t"
            "v_cndmask_b32 %[w], %[w3], %[w], vcc\n\t"
            "v_add_u32 %[n], 1, %[n]\n\t"
            "v_mad_u32_u24 %[ls], %[w], 12, 1\n\t"
            "v_cmp_lt_u32 vcc, %[pw], %[stop]\n\t"
            "s_and_b64 exec, exec, vcc\n\t"
            "s_cbranch_scc0 9f\n\t"
            "v_cmp_eq_u32 vcc, -1, %[bits]\n\t"
            "s_sub_u32 %[cnt], %[cnt], 1\n\t"
            "s_cbranch_scc0 1b\n"
            "9:\n\t"
            "s_mov_b64 exec, %[ex]\n"
            : [pw] "+v"(pw), [w] "+v"(w), [n] "+v"(n), [ls] "+v"(ls), [ex] "=&s"(t_ex), [a] "=&v"(t_a), [bits] "=&v"(t_bits),
              [w3] "=&v"(t_w3), [wa] "=&v"(t_wa), [wb] "=&v"(t_wb), [lx] "=&v"(t_hx), [t] "=&v"(t_t), [t1] "=&v"(t_li), [t2] "=&v"(pos),
              [cnt] "+s"(cnt)
            : [stop] "v"(stop), [c90] "s"(c90), [c132] "s"(c132)
            : "vcc", "scc", "memory", "v62", "v63");
        pos = pw;
    } else if (VARIANT == 4) {                              // two chains per lane (rows of lane l and of lane l ^ 32), software-pipelined, interleaved
        uint32_t pwa = 8u * rowb + pos, pwb = 8u * (uint32_t)(uintptr_t)(&win[wave][(lane ^ 32u) * kRow]) + pos + 5u;
        uint32_t wa_ = w, wb_ = w + 1u, na = 0u, nbb = 0u, lsa = 1u + 12u * wa_, lsb = 1u + 12u * wb_;
        const uint32_t stop = 0xF0000000u, c90 = 90u, c132 = 132u;
        uint32_t a1, b1, w31, wa1, wb1, lx1, t1, u1, v1, a2, b2, w32, wa2, wb2, lx2, t2, u2, v2;
        asm volatile(
            "s_mov_b64 %[ex], exec\n\t"
            "v_lshrrev_b32 %[a1], 3, %[pwa]\n\t"
            "v_and_b32 %[a1], 0xffc, %[a1]\n\t"
            "ds_read2_b32 v[62:63], %[a1] offset1:1\n\t"
            "v_lshrrev_b32 %[a2], 3, %[pwb]\n\t"
            "v_and_b32 %[a2], 0xffc, %[a2]\n\t"
            "ds_read2_b32 v[60:61], %[a2] offset1:1\n"
            "1:\n\t"
            "s_waitcnt lgkmcnt(1)\n\t"
            "v_alignbit_b32 %[b1], v63, v62, %[pwa]\n\t"
            "v_bfe_u32 %[w31], %[b1], 1, 3\n\t"
            "v_bfe_u32 %[wa1], %[b1], 4, 2\n\t"
            "v_bfe_u32 %[wb1], %[b1], 6, 6\n\t"
            "v_and_b32 %[t1], 1, %[b1]\n\t"
            "s_waitcnt lgkmcnt(0)\n\t"
            "v_alignbit_b32 %[b2], v61, v60, %[pwb]\n\t"
            "v_bfe_u32 %[w32], %[b2], 1, 3\n\t"
            "v_bfe_u32 %[wa2], %[b2], 4, 2\n\t"
            "v_bfe_u32 %[wb2], %[b2], 6, 6\n\t"
            "v_and_b32 %[t2], 1, %[b2]\n\t"
            "v_mad_u32_u24 %[lx1], %[w31], 12, 4\n\t"
            "v_mad_u32_u24 %[u1], %[wa1], 12, %[c90]\n\t"
            "v_mad_u32_u24 %[v1], %[wb1], 12, %[c132]\n\t"
            "v_mad_u32_u24 %[lx2], %[w32], 12, 4\n\t"
            "v_mad_u32_u24 %[u2], %[wa2], 12, %[c90]\n\t"
            "v_mad_u32_u24 %[v2], %[wb2], 12, %[c132]\n\t"
            "v_cmp_eq_u32 vcc, 3, %[wa1]\n\t"
            "v_cndmask_b32 %[u1], %[u1], %[v1], vcc\n\t"
            "v_add_u32 %[wa1], 7, %[wa1]\n\t"
            "v_add_u32 %[wb1], 10, %[wb1]\n\t"
            "v_cndmask_b32 %[wa1], %[wa1], %[wb1], vcc\n\t"
            "v_cmp_eq_u32 vcc, 3, %[wa2]\n\t"
            "v_cndmask_b32 %[u2], %[u2], %[v2], vcc\n\t"
            "v_add_u32 %[wa2], 7, %[wa2]\n\t"
            "v_add_u32 %[wb2], 10, %[wb2]\n\t"
            "v_cndmask_b32 %[wa2], %[wa2], %[wb2], vcc\n\t"
            "v_cmp_eq_u32 vcc, 7, %[w31]\n\t"
            "v_cndmask_b32 %[lx1], %[lx1], %[u1], vcc\n\t"
            "v_cndmask_b32 %[w31], %[w31], %[wa1], vcc\n\t"
            "v_cmp_eq_u32 vcc, 7, %[w32]\n\t"
            "v_cndmask_b32 %[lx2], %[lx2], %[u2], vcc\n\t"
            "v_cndmask_b32 %[w32], %[w32], %[wa2], vcc\n\t"
            "v_cmp_eq_u32 vcc, 1, %[t1]\n\t"
            "v_cndmask_b32 %[lx1], %[lx1], %[lsa], vcc\n\t"
            "v_add_u32 %[pwa], %[pwa], %[lx1]\n\t"
            "v_lshrrev_b32 %[a1], 3, %[pwa]\n\t"
            "v_and_b32 %[a1], 0xffc, %[a1]\n\t"
            "ds_read2_b32 v[62:63], %[a1] offset1:1\n\t"
            "v_cndmask_b32 %[wA], %[w31], %[wA], vcc\n\t"
            "v_cmp_eq_u32 vcc, 1, %[t2]\n\t"
            "v_cndmask_b32 %[lx2], %[lx2], %[lsb], vcc\n\t"
            "v_add_u32 %[pwb], %[pwb], %[lx2]\n\t"
            "v_lshrrev_b32 %[a2], 3, %[pwb]\n\t"
            "v_and_b32 %[a2], 0xffc, %[a2]\n\t"
            "ds_read2_b32 v[60:61], %[a2] offset1:1\n\t"
            "v_cndmask_b32 %[wB], %[w32], %[wB], vcc\n\t"
            "v_add_u32 %[na], 1, %[na]\n\t"
            "v_add_u32 %[nb], 1, %[nb]\n\t"
            "v_mad_u32_u24 %[lsa], %[wA], 12, 1\n\t"
            "v_mad_u32_u24 %[lsb], %[wB], 12, 1\n\t"
            "v_cmp_lt_u32 vcc, %[pwa], %[stop]\n\t"
            "s_and_b64 exec, exec, vcc\n\t"
            "v_cmp_lt_u32 vcc, %[pwb], %[stop]\n\t"
            "s_and_b64 exec, exec, vcc\n\t"
            "s_cbranch_scc0 9f\n\t"
            "s_sub_u32 %[cnt], %[cnt], 1\n\t"
            "s_cbranch_scc0 1b\n"
            "9:\n\t"
            "s_waitcnt lgkmcnt(0)\n\t"
            "s_mov_b64 exec, %[ex]\n"
            : [pwa] "+v"(pwa), [pwb] "+v"(pwb), [wA] "+v"(wa_), [wB] "+v"(wb_), [na] "+v"(na), [nb] "+v"(nbb), [lsa] "+v"(lsa), [lsb] "+v"(lsb),
              [ex] "=&s"(t_ex), [a1] "=&v"(a1), [b1] "=&v"(b1), [w31] "=&v"(w31), [wa1] "=&v"(wa1), [wb1] "=&v"(wb1), [lx1] "=&v"(lx1),
              [t1] "=&v"(t1), [u1] "=&v"(u1), [v1] "=&v"(v1), [a2] "=&v"(a2), [b2] "=&v"(b2), [w32] "=&v"(w32), [wa2] "=&v"(wa2),
              [wb2] "=&v"(wb2), [lx2] "=&v"(lx2), [t2] "=&v"(t2), [u2] "=&v"(u2), [v2] "=&v"(v2), [cnt] "+s"(cnt)
            : [stop] "v"(stop), [c90] "s"(c90), [c132] "s"(c132)
            : "vcc", "scc", "memory", "v60", "v61", "v62", "v63");
        pos = pwa + pwb; n = na + nbb; w = wa_ + wb_;
    } else {                                                // no lane masks, no scalar work besides the loop counter: selects only
        asm volatile(
            "1:\n\t"
            "v_subrev_u32 %[li], %[w0c], %[pos]\n\t"
            "v_and_b32 %[li], 0x3ff, %[li]\n\t"
            "v_lshrrev_b32 %[a], 5, %[li]\n\t"
            "v_lshl_add_u32 %[a], %[a], 2, %[rowb]\n\t"
            "ds_read2_b32 v[62:63], %[a] offset1:1\n\t"
            "s_waitcnt lgkmcnt(0)\n\t"
            "v_alignbit_b32 %[bits], v63, v62, %[li]\n\t"
            "v_bfe_u32 %[w3], %[bits], 1, 3\n\t"
            "v_bfe_u32 %[wa], %[bits], 4, 2\n\t"
            "v_bfe_u32 %[wb], %[bits], 6, 6\n\t"
            "v_cmp_eq_u32 %[sa], 7, %[w3]\n\t"
            "v_cmp_eq_u32 vcc, 3, %[wa]\n\t"
            "v_add_u32 %[wa], 7, %[wa]\n\t"
            "v_add_u32 %[wb], 10, %[wb]\n\t"
            "v_cndmask_b32 %[wa], %[wa], %[wb], vcc\n\t"
            "v_cndmask_b32 %[hx], 6, 12, vcc\n\t"
            "v_cndmask_b32 %[wa], %[w3], %[wa], %[sa]\n\t"
            "v_cndmask_b32 %[hx], 4, %[hx], %[sa]\n\t"
            "v_and_b32 %[t], 1, %[bits]\n\t"
            "v_cmp_eq_u32 vcc, 1, %[t]\n\t"
            "v_cndmask_b32 %[w], %[wa], %[w], vcc\n\t"
            "v_cndmask_b32 %[hx], %[hx], 1, vcc\n\t"
            "v_min_u32 %[w], %[w], %[maxw]\n\t"
            "v_mad_u32_u24 %[t], %[w], 12, %[hx]\n\t"
            "v_add_u32 %[pos], %[pos], %[t]\n\t"
            "v_add_u32 %[n], 1, %[n]\n\t"
            "s_sub_u32 %[cnt], %[cnt], 1\n\t"
            "s_cbranch_scc0 1b\n"
            : [pos] "+v"(pos), [w] "+v"(w), [n] "+v"(n), [sa] "=&s"(t_sa), [li] "=&v"(t_li), [a] "=&v"(t_a), [bits] "=&v"(t_bits),
              [w3] "=&v"(t_w3), [wa] "=&v"(t_wa), [wb] "=&v"(t_wb), [hx] "=&v"(t_hx), [t] "=&v"(t_t), [cnt] "+s"(cnt)
            : [w0c] "s"(w0c), [rowb] "v"(rowb), [maxw] "v"(maxw)
            : "vcc", "scc", "memory", "v62", "v63");
    }
    const uint64_t t1 = clock64();
    const uint64_t r1 = wall_clock64();
    if (lane == 0) { out[2 * (blockIdx.x * 4 + wave)] = t1 - t0; out[2 * (blockIdx.x * 4 + wave) + 1] = (r1 - r0) + ((uint64_t)((pos + n + w) & 1u) << 60); }
}

int main() {
    const uint32_t steps = 4000;
    const int max_blocks = 256 * 8;
    std::vector<uint32_t> h((size_t)max_blocks * 256 * kRow);
    uint32_t x = 12345;
    for (auto& v : h) { x ^= x << 13; x ^= x >> 17; x ^= x << 5; v = x; }
    uint32_t* seed; uint64_t* out;
    hipMalloc(&seed, h.size() * 4); hipMalloc(&out, max_blocks * 4 * 16);
    hipMemcpy(seed, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    std::vector<uint64_t> r(max_blocks * 8);
    for (int variant = 0; variant < 5; ++variant)
        for (int per_cu = 1; per_cu <= 4; per_cu *= 2) {      // workgroups of 4 waves per CU = waves per SIMD
            const int blocks = 256 * per_cu;
            for (int rep = 0; rep < 2; ++rep) {
                if (variant == 0) hipLaunchKernelGGL(k_step<0>, dim3(blocks), dim3(256), 0, 0, seed, steps, out);
                if (variant == 1) hipLaunchKernelGGL(k_step<1>, dim3(blocks), dim3(256), 0, 0, seed, steps, out);
                if (variant == 4) hipLaunchKernelGGL(k_step<4>, dim3(blocks), dim3(256), 0, 0, seed, steps, out);
                if (variant == 3) hipLaunchKernelGGL(k_step<3>, dim3(blocks), dim3(256), 0, 0, seed, steps, out);
                if (variant == 2) hipLaunchKernelGGL(k_step<2>, dim3(blocks), dim3(256), 0, 0, seed, steps, out);
                hipDeviceSynchronize();
            }
            hipMemcpy(r.data(), out, blocks * 4 * 16, hipMemcpyDeviceToHost);
            double sum = 0, rs = 0; uint64_t mx = 0;
            for (int i = 0; i < blocks * 4; ++i) { sum += r[2 * i]; rs += r[2 * i + 1] & 0xFFFFFFFFFFull; mx = mx > r[2 * i] ? mx : r[2 * i]; }
            printf("variant %d (%s) waves/SIMD %d: %.1f ns per step (100 MHz counter), %.1f clock64 ticks per step (mean), %.1f (slowest wave)\n", variant,
                   variant == 0 ? "full step" : variant == 1 ? "no LDS read" : variant == 2 ? "selects only, no lane masks" : variant == 3 ? "software-pipelined" : "two chains per lane, per DOUBLE step", per_cu, rs / (blocks * 4) / steps * 10.0,
                   sum / (blocks * 4) / steps, (double)mx / steps);
        }
    return 0;
}
