"""Host-side model of the DATA PATH of ``tap_slab_kernel`` (daam_amd/csrc/daam_tap_slab.hip): which instruction of which wave fetches
which 16-byte piece into which LDS slot (the per-lane offsets shared by instructions 10 apart, the scalar 16-row steps, the extra
instructions of waves 0..3 / 4..7, the half-size tiles), the swizzle, the operand reads of every wave role (head_dim 40: wave = head,
two pixel groups; 80: wave = (head, group); 160: four waves), the zero piece behind the tail k-step -- lane by lane, with the documented
operand layout of ``v_mfma_f32_16x16x32_f16``, checked against ``Q K^T`` for every head of a slab.

Every formula below is the kernel's, transcribed (names kept): a wrong offset / slot / role shows up as a wrong logit or as a NaN (LDS is
poisoned before the fetches; guard halves behind the tensors are NaN).  What it cannot check: instruction-level behaviour (DMA completion,
barriers, hazards) -- that is what tests/test_gpu_slab.py (bit-identity with the other kernels on the chip) and tools/exp/soak.py are for.
Run: ``python tools/emulate_tap_slab.py``; ``tests/test_host_logic.py::test_tap_slab_data_path_model`` runs the same cases.
"""
from __future__ import annotations

import numpy as np

K_TOK = 77
SLAB_BYTES = 640
SLOTS = 40
SLAB_PX = 32
K_BYTES = 80 * SLAB_BYTES
Q_OFF = K_BYTES
Q_BYTES = SLAB_PX * SLAB_BYTES
ZERO_OFF = Q_OFF + Q_BYTES
K_INSTR = (K_TOK * SLOTS + 63) // 64        # 49
WAVES = 8


def emulate_workgroup(q, k, head_dim, tp, px_end, p0, q_sp, k_st, q_off, k_off):
    """One workgroup (slab = the 640 bytes from q_off / k_off on, pixels p0 .. p0 + tp), one denoising step.  ``q`` / ``k``: flat fp16
    arrays; strides / offsets in elements.  Returns logits [heads of the slab, tp pixels, 80 token slots] (q . k, unscaled, f32)."""
    qb, kb = q.view(np.uint8), k.view(np.uint8)
    lds = np.full(ZERO_OFF + 16, 0xFF, np.uint8)        # poison: 0xFFFF halves are NaN
    lds[ZERO_OFF:ZERO_OFF + 16] = 0                     # the zero piece
    lds[K_TOK * SLAB_BYTES:K_BYTES] = 0                 # K rows 77..79: cleared once at kernel start
    pph = head_dim // 8
    nh = SLOTS // pph
    nks = (pph + 3) // 4
    g_per_wave = tp // 16 if nh == 8 else 1
    items = nh * (tp // 16 // g_per_wave)
    lanes = np.arange(64)

    def lane_off(i, lane, row_stride, row0, row_max):
        sigma = 64 * i + lane
        row, t = divmod(sigma, SLOTS)
        return (row0 + min(row, row_max)) * row_stride * 2 + ((t ^ ((row >> 1) & 7)) << 4)

    k_base, q_base = k_off * 2, q_off * 2
    k16 = 16 * k_st * 2
    q16 = 16 * q_sp * 2 if px_end - p0 >= SLAB_PX else 0
    for wave in range(WAVES):
        wx = 8 + (wave & 1)
        for lane in lanes:
            kdA = lane_off(wave, lane, k_st, 0, K_TOK - 1)
            kdB = lane_off(wx, lane, k_st, 0, K_TOK - 1)
            qdA = lane_off(wave, lane, q_sp, p0, 15)
            xd = lane_off(wx, lane, q_sp, p0, 15) if wave < 4 else lane_off(K_INSTR - 1, lane, k_st, 0, K_TOK - 1)
            # dma_k
            for m in range(5):
                src = kdA + k_base + m * k16
                dst = (wave + 10 * m) * 1024 + lane * 16
                lds[dst:dst + 16] = kb[src:src + 16]
            src = kdB + k_base + (wave >> 1) * k16
            dst = (wx + 10 * (wave >> 1)) * 1024 + lane * 16
            lds[dst:dst + 16] = kb[src:src + 16]
            if wave >= 4:
                src = xd + k_base
                dst = (K_INSTR - 1) * 1024 + lane * 16
                lds[dst:dst + 16] = kb[src:src + 16]
            # dma_q
            src = qdA + q_base
            dst = Q_OFF + wave * 1024 + lane * 16
            lds[dst:dst + 16] = qb[src:src + 16]
            if tp == SLAB_PX:
                src = qdA + q_base + q16
                dst = Q_OFF + (wave + 10) * 1024 + lane * 16
                lds[dst:dst + 16] = qb[src:src + 16]
            if wave < tp // 8:
                src = xd + q_base + (wave >> 1) * q16
                dst = Q_OFF + (wx + 10 * (wave >> 1)) * 1024 + lane * 16
                lds[dst:dst + 16] = qb[src:src + 16]

    def rd(off):
        return lds[off:off + 16].view(np.float16).astype(np.float32)

    out = np.full((nh, tp, 80), np.nan, np.float32)
    for wave in range(WAVES):
        if not (items == WAVES or wave < items):
            continue
        head, grp = wave % nh, (0 if g_per_wave == 2 else wave // nh)
        acc = np.zeros((g_per_wave, 5, 64, 4), np.float32)
        for ks in range(nks):
            a = np.zeros((5, 64, 8), np.float32)
            bq = np.zeros((g_per_wave, 64, 8), np.float32)
            for lane in lanes:
                j, h = lane & 15, lane >> 4
                pi = 4 * ks + h
                valid = pi < pph
                p = head * pph + (pi if valid else 4 * ks)
                f_k = j * SLAB_BYTES + ((p ^ ((j >> 1) & 7)) << 4)
                f_q = (Q_OFF + 16 * grp * SLAB_BYTES + f_k) if valid else ZERO_OFF
                for g in range(g_per_wave):
                    off = f_q if g == 0 else ((f_q + 16 * SLAB_BYTES) if valid else ZERO_OFF)
                    bq[g, lane] = rd(off)
                for mt in range(5):
                    a[mt, lane] = rd(mt * 16 * SLAB_BYTES + f_k)
            for mt in range(5):
                A = np.zeros((16, 32), np.float32)
                for lane in lanes:
                    A[lane & 15, 8 * (lane >> 4):8 * (lane >> 4) + 8] = a[mt, lane]
                for g in range(g_per_wave):
                    B = np.zeros((16, 32), np.float32)
                    for lane in lanes:
                        B[lane & 15, 8 * (lane >> 4):8 * (lane >> 4) + 8] = bq[g, lane]
                    D = A @ B.T                                       # [token row, pixel column]
                    for lane in lanes:
                        acc[g, mt, lane] += D[4 * (lane >> 4):4 * (lane >> 4) + 4, lane & 15]
        for lane in lanes:
            j, h = lane & 15, lane >> 4
            for g in range(g_per_wave):
                for mt in range(5):
                    out[head, 16 * (grp + g) + j, 16 * mt + 4 * h:16 * mt + 4 * h + 4] = acc[g, mt, lane]
    return out


def check(head_dim, hw, tp=SLAB_PX, px_begin=0, px_end=None, tile=0, heads=None, batch=2, seed=0, verbose=True):
    """The LAST slab of the last batch (pieces past the slab would run off the row / the tensor): every head of it against Q K^T."""
    rng = np.random.default_rng(seed)
    nh = SLOTS // (head_dim // 8)
    heads = heads or 2 * nh
    px_end = hw if px_end is None else px_end
    C = heads * head_dim
    q = (rng.standard_normal((batch, hw, C)) * 0.5).astype(np.float16)
    k = (rng.standard_normal((batch, K_TOK, C)) * 0.5).astype(np.float16)
    b, hd0 = batch - 1, heads - nh
    q_off = b * hw * C + hd0 * head_dim
    k_off = b * K_TOK * C + hd0 * head_dim
    qf = np.concatenate([q.reshape(-1), np.full(512, np.nan, np.float16)])
    kf = np.concatenate([k.reshape(-1), np.full(512, np.nan, np.float16)])
    p0 = px_begin + tile * tp
    got = emulate_workgroup(qf, kf, head_dim, tp, px_end, p0, C, C, q_off, k_off)
    n_px = min(tp, px_end - p0)
    err = 0.0
    for hh in range(nh):
        qs = q[b, p0:p0 + n_px, (hd0 + hh) * head_dim:(hd0 + hh + 1) * head_dim].astype(np.float32)
        ks = k[b, :, (hd0 + hh) * head_dim:(hd0 + hh + 1) * head_dim].astype(np.float32)
        err = max(err, float(np.abs(got[hh, :n_px, :K_TOK] - qs @ ks.T).max()))
    finite = bool(np.isfinite(got[:, :n_px]).all())                 # incl. the padding tokens 77..79: finite filler, masked by the chain start
    if verbose:
        print(f'head_dim {head_dim:3d} hw {hw:5d} tile {tile} of {tp} px from {px_begin}: max |logit - q.k| = {err:.2e}, finite: {finite}')
    return err, finite


CASES = [dict(head_dim=40, hw=256), dict(head_dim=40, hw=256, tile=7), dict(head_dim=80, hw=64, tile=1), dict(head_dim=160, hw=64),
         dict(head_dim=160, hw=64, tile=1), dict(head_dim=80, hw=144, tile=4),                # hw = 144: the last tile's second half is outside
         dict(head_dim=40, hw=256, tp=16, px_begin=192, tile=3), dict(head_dim=40, hw=16), dict(head_dim=40, hw=4096, tile=127, heads=16)]

if __name__ == '__main__':
    for c in CASES:
        e, fin = check(**c)
        assert e < 2e-3 and fin, c
    print('ok')
