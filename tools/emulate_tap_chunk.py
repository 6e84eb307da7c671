"""Host-side model of the DATA PATH of ``tap_chunk_kernel`` (daam_amd/csrc/daam_tap_chunk.hip): the index arithmetic of its
LDS-DMA fetches (swizzled source pieces, clamped rows, partial last chunk), its operand reads, the Q-operand masks and the k-step
skip rule, lane by lane, with the documented operand layout of ``v_mfma_f32_16x16x32_f16`` -- checked against ``Q K^T``.

The kernel was written in a session without a GPU; this model is what could be checked there: every formula below is the
kernel's, transcribed (names kept), so a wrong offset / mask / chunk rule shows up as a wrong logit here.  What it cannot check:
the instruction-level behaviour (DMA completion, barriers, hazards).  Run: ``python tools/emulate_tap_chunk.py``; the CPU test
``tests/test_host_logic.py::test_tap_chunk_data_path_model`` runs the same cases.
"""
from __future__ import annotations

import numpy as np

K_TOK = 77
CK_ROW = 128
CK_KBUF = 80 * CK_ROW
CK_QTILE = 32 * CK_ROW
CK_QOFF = 2 * CK_KBUF
PIXELS = 128


def ck_swz(row, chunk):
    return ((chunk ^ ((row >> 1) & 7)) << 4)


def emulate_workgroup(q, k, head_dim, hw, p0, q_sp, k_st, q_off, k_off):
    """One workgroup, one denoising step.  ``q`` / ``k``: flat fp16 arrays (the step's tensors); strides in elements.
    Returns logits [128 pixels of the tile, 80 token slots] as f32 (q . k, unscaled)."""
    qb = q.view(np.uint8)
    kb_g = k.view(np.uint8)
    lds = np.zeros(CK_QOFF + 4 * CK_QTILE, np.uint8)
    lds[:] = 0xFF                                       # poison: 0xFFFF halves are NaN -- anything read before written shows up
    d = head_dim
    n_ch = (d + 63) >> 6
    vc = (d - 64 * (n_ch - 1)) >> 3
    last_partial = vc < 8
    out = np.zeros((PIXELS, 80), np.float32)
    acc = np.zeros((4, 64, 2, 5, 4), np.float32)        # wave, lane, group, mt, r
    lanes = np.arange(64)

    def dma(c, buf):
        cb = c * 128
        lastp = last_partial and c == n_ch - 1
        for wave in range(4):
            for j2 in range(3):
                blk = 4 * j2 + wave
                if blk >= 10:
                    continue
                for lane in lanes:
                    rowu = 8 * blk + (lane >> 3)
                    row = min(rowu, K_TOK - 1)
                    ch = (lane & 7) ^ ((rowu >> 1) & 7)
                    kd_full = (row * k_st + ch * 8) * 2
                    kd_last = (row * k_st + (ch * 8 if ch < vc else 0)) * 2
                    src = (kd_last if lastp else kd_full) + k_off * 2 + cb
                    dst = buf * CK_KBUF + blk * 1024 + lane * 16
                    lds[dst:dst + 16] = kb_g[src:src + 16]
            q_rows_in = hw - (p0 + wave * 32)
            for i in range(4):
                q_s = i * 8 * q_sp * 2 if 8 * i < q_rows_in else 0
                par = i & 1
                for lane in lanes:
                    ch = (lane & 7) ^ ((4 * par + (lane >> 4)) & 7)
                    px = p0 + wave * 32 + (lane >> 3)
                    rowb = (q_off + min(px, hw - 1) * q_sp) * 2
                    qd_full = rowb + ch * 16
                    qd_last = rowb + (ch * 16 if ch < vc else 0)
                    src = (qd_last if lastp else qd_full) + q_s + cb
                    dst = CK_QOFF + wave * CK_QTILE + i * 1024 + lane * 16
                    lds[dst:dst + 16] = qb[src:src + 16]

    def rd(off):
        return lds[off:off + 16].view(np.float16).astype(np.float32)

    def mfma_sub(buf, first, partial):
        for wave in range(4):
            qt = CK_QOFF + wave * CK_QTILE
            kbase = buf * CK_KBUF
            for ks in range(2):
                if ks == 1 and partial and not vc > 4:
                    continue
                a = np.zeros((5, 64, 8), np.float32)
                bq = np.zeros((2, 64, 8), np.float32)
                for lane in lanes:
                    jj, h = lane & 15, lane >> 4
                    f_rd = jj * CK_ROW + ck_swz(jj, h)
                    off = f_rd ^ 64 if ks else f_rd
                    for g in range(2):
                        v = rd(qt + g * 16 * CK_ROW + off)
                        if partial and ((ks == 0 and vc < 4 and h >= vc) or (ks == 1 and 4 + h >= vc)):
                            v = np.zeros(8, np.float32)
                        bq[g, lane] = v
                    for mt in range(5):
                        a[mt, lane] = rd(kbase + mt * 16 * CK_ROW + off)
                # v_mfma_f32_16x16x32_f16: A lane l = row l & 15, k 8 (l >> 4) .. + 7; B lane l = column l & 15, same k;
                # D lane l = column l & 15, rows 4 (l >> 4) + r
                for mt in range(5):
                    A = np.zeros((16, 32), np.float32)
                    for lane in lanes:
                        A[lane & 15, 8 * (lane >> 4):8 * (lane >> 4) + 8] = a[mt, lane]
                    for g in range(2):
                        B = np.zeros((16, 32), np.float32)
                        for lane in lanes:
                            B[lane & 15, 8 * (lane >> 4):8 * (lane >> 4) + 8] = bq[g, lane]
                        D = A @ B.T                                   # [token row, pixel column]
                        for lane in lanes:
                            for r in range(4):
                                val = D[4 * (lane >> 4) + r, lane & 15]
                                if first and ks == 0:
                                    acc[wave, lane, g, mt, r] = val
                                else:
                                    acc[wave, lane, g, mt, r] += val

    dma(0, 0)
    buf = 0
    mfma_sub(buf, True, last_partial and n_ch == 1)
    buf ^= 1
    if n_ch > 1:
        dma(1, buf)
    for c in range(1, n_ch):
        mfma_sub(buf, False, last_partial and c == n_ch - 1)
        buf ^= 1
        if c + 1 < n_ch:
            dma(c + 1, buf)
    for wave in range(4):
        for lane in lanes:
            jj, h = lane & 15, lane >> 4
            for g in range(2):
                for mt in range(5):
                    for r in range(4):
                        out[wave * 32 + 16 * g + jj, 16 * mt + 4 * h + r] = acc[wave, lane, g, mt, r]
    return out


def check(head_dim, hw, heads=2, batch=2, p0=0, seed=0, verbose=True):
    rng = np.random.default_rng(seed)
    C = heads * head_dim
    q = (rng.standard_normal((batch, hw, C)) * 0.5).astype(np.float16)
    k = (rng.standard_normal((batch, K_TOK, C)) * 0.5).astype(np.float16)
    b, hd = batch - 1, heads - 1                                   # the LAST head of the last batch: pieces past head_dim would
    q_off = b * hw * C + hd * head_dim                             # run off the row / the tensor if they were not clamped
    k_off = b * K_TOK * C + hd * head_dim
    # guard: append NaN halves past the end so that an unclamped read of "the next head" is seen
    qf = np.concatenate([q.reshape(-1), np.full(256, np.nan, np.float16)])
    kf = np.concatenate([k.reshape(-1), np.full(256, np.nan, np.float16)])
    got = emulate_workgroup(qf, kf, head_dim, hw, p0, C, C, q_off, k_off)
    qs = q[b, :, hd * head_dim:(hd + 1) * head_dim].astype(np.float32)
    ks = k[b, :, hd * head_dim:(hd + 1) * head_dim].astype(np.float32)
    ref = qs @ ks.T                                                # [hw, 77]
    n_px = min(PIXELS, hw - p0)
    err = np.abs(got[:n_px, :K_TOK] - ref[p0:p0 + n_px]).max()
    pad_finite = np.isfinite(got).all()
    if verbose:
        print(f'head_dim {head_dim:3d} hw {hw:5d} p0 {p0:5d}: max |logit - q.k| = {err:.2e}, padded slots finite: {pad_finite}')
    return err, pad_finite


CASES = [(8, 64, 0), (40, 256, 128), (64, 128, 0), (72, 64, 0), (80, 256, 0), (96, 128, 0), (120, 64, 0), (128, 128, 0),
         (160, 256, 128), (200, 64, 0), (256, 128, 0), (40, 72, 0), (160, 200, 128)]

if __name__ == '__main__':
    for d, hw, p0 in CASES:
        e, fin = check(d, hw, p0=p0)
        assert e < 2e-3 and fin, (d, hw, p0)
    print('ok')
