#!/usr/bin/env python
"""Generates daam_amd/csrc/daam_finalize_pipe_{prefill,asm}_{r16,bf16,f32}.inc (one schedule per dtype of the running sums: fp16 planes = r16,
bf16 planes -- which share the fp16 prefill --, f32 planes): the software-pipelined main loop of the
x2 (32 -> 64) finalize on the matrix cores as hand-scheduled inline asm (fixed physical registers, counted waits, LDS-DMA plane
ring): statement 1 starts the ring's first R planes, statement 2 is the pipeline (the kernel runs its same-size keys between the
two, under the latency of the prefill).

    python tools/gen_fin_pipe.py            # rewrites the .inc (committed next to this script's output)

Why a generator: the loop is a two-deep software pipeline over planes with double-buffered registers -- per iteration i
    C(i)    pass 2 of plane i:   8 MFMA 32x32x16  (o[pi] = Wy x (hi + lo of T(i)) + acc[pi]),  pi = i & 1
    D(i-1)  clamp + accumulate of plane i-1: 32 v_max_i32  (acc[1-pi] = max(o[1-pi], acc[1-pi]))
    B(i+1)  hi / lo split of T(i+1): 8 v_cvt_pk + 16 v_fma_mix + 8 v_cvt_pk -> B[1-pi]
    A(i+2)  pass 1 of plane i+2: 2 MFMA (T = P x Wx^T), plane from the LDS ring
    + DMA of plane i+1+R into the ring slot plane i+1 left, the next key pointer by s_load
-- and every VALU instruction sits in a fixed gap behind an MFMA that does not depend on it (6-7 per gap: the measured issue
budget of a gfx950 SIMD).  hipcc cannot be talked into this schedule (it re-serialises the stages and
ties vdst = srcC), and an asm statement gets no hazard padding from it: the distances below ARE the hazard handling
(cdna_hip_programming.md section 5.7):
  * MFMA result -> VALU read: >= 2 later MFMAs of the in-order matrix pipe issued in between (>= 64 cycles; 12 states needed);
  * VALU write -> MFMA operand: >= 2 instructions in between;  T is rewritten by A(i+2) only after every B(i+1) instruction;
  * the ring belongs to the WORKGROUP: each of its two waves (the two 32-column output halves) fetches one 1 KiB half of every
    plane and both read all of it -- a plane crosses L2 -> LDS once (with wave-private rings the duplicate fetches doubled the
    kernel's traffic to 6.7 TB/s and bounded it).  LDS-DMA -> ds_read: each wave's counted vmcnt, THEN s_barrier (the partner's
    half), then the reads;  ds_read -> MFMA: lgkmcnt(0);
  * ring slot reuse: the DMA into slot (i+1) % R is issued after this iteration's barrier, which both waves pass only after the
    lgkmcnt(0) that retired their reads of plane i+1.
"""
import os

import sys

# Plane dtype of the generated variant (round 6): 'f16' (the original, described above), 'bf16' (pass 1 on v_mfma_f32_32x32x16_bf16: the plane is
# its A operand as it is, the tap matrix goes in as two bf16 matrices W' + E with a third MFMA for E -- see PMIX below; everything behind pass 1
# is the f16 variant's), 'f32' (4 KiB planes: every lane reads its A pieces as 16
# floats and splits them into an fp16 hi + lo pair like pass 2 splits T -- 32 more VALU instructions and 2 more MFMAs per plane; ring of
# 8 planes = the same 32 KiB; two LDS-DMA instructions per plane and wave; the 16-byte pieces of a 128-byte plane row are XOR-swizzled
# with (row >> 1) & 7 on the SOURCE address so that the ds_read_b128 of 32 rows x one piece column are conflict-free).
# main() writes all three; DAAM_PIPE_DT selects one (experiments).
DT = 'f16'
R = 16               # ring slots (planes) per workgroup: 16 x 2 KiB or 8 x 4 KiB (the pointers of the first R planes come in by s_load_dwordx16)
# timing experiments (results are wrong): 1 no DMA / key loads in the loop, 2 no barrier, 3 no plane reads / vmcnt waits,
# 4 no VALU (MFMA stream only), 5 no MFMA (VALU stream only)
ABLATE = int(os.environ.get('DAAM_PIPE_ABLATE', '0'))
# MFMA / VALU arrangement of a steady iteration: 0 one MFMA per gap (two accumulator chains alternating), 1 the three dependent
# chains back to back (c0 x4 | c1 x4 | a x2) with the VALU in three blocks behind them, 2 = 1 + s_setprio around the chains,
# 3 half chains (pairs)
SCHED = int(os.environ.get('DAAM_PIPE_SCHED', '0'))
# experiment (tools/exp): time-sliced issue priority.  The two waves that share a SIMD come from different workgroups and the older
# one wins the arbitration until it is done; with FAIR = b each wave raises its priority while bit b of the shader clock equals the
# parity of its wave slot (input s46), so the two take turns.  0 = off (the shipped schedule).
FAIR = int(os.environ.get('DAAM_PIPE_FAIR', '0'))
S_TIME, S_TMP, S_PRIO_ID = '50:51', 52, 46
_label = [0]
NT = ' nt' if os.environ.get('DAAM_PIPE_NT') else ''     # cache policy of the plane fetches (every plane is read once)
SLOT = 2048
LOADS = 1            # LDS-DMA instructions per plane and wave (f32: 2, the second 16 rows = 2048 bytes further in both address spaces)
# ---- register map --------------------------------------------------------------------------------------------------
WX = (0, 4)               # B operands of pass 1 (4 VGPRs each)
WY = {(0, 0): 8, (0, 1): 12, (1, 0): 16, (1, 1): 20}    # wy[t][ks]: A operands of pass 2
P = (24, 28)              # plane pieces (A operands of pass 1)
T = 32                    # 16
def BSET(s): return 48 + 16 * s          # bhi0 +0, bhi1 +4, blo0 +8, blo1 +12
def OSET(s): return 80 + 32 * s          # o0 +0, o1 +16
def ASET(s): return 144 + 32 * s         # acc0 +0, acc1 +16
GOFF, LDS_RD, LDS_TMP = 208, 210, 211          # GOFF: this wave's half of a plane (tok * 2048 + nt * 1024 + lane * 16)
# f32 planes: the raw pieces, the lo halves and four read addresses (piece columns 2g, 2g + 1, 4 + 2g, 5 + 2g of row n, swizzled)
PRAW, PLO, LDS_RD4, LDS_TMP4 = 220, (212, 216), 236, 240
# bf16 planes: the 32 -> 64 tap matrix has ONE entry per border that is no bf16 number (three taps clamped onto the border column add up to
# 283/256: nine significant bits); the host splits W = W' + E (both bf16-exact, E = 1/256 at [0][0] and [63][31]) and pass 1 gets a third MFMA
# T += PMIX x E^T, whose A operand PMIX is columns 0..7 (lanes g = 0: their k-step-0 piece) and 24..31 (lanes g = 1: their k-step-1 piece)
# of the plane -- four v_cndmask per plane, no extra LDS traffic
PMIX, WXE, S_LOW32 = 212, 216, '52:53'
S_KEYS = '36:37'
S_KOFF, S_TRIPS, S_RING, S_RDSLOT, S_DMASLOT, S_M0SAVE = 38, 39, 42, 43, 44, 47
S_BASE = '40:41'
S_PRE = 48                # s[48:63] (R = 16: .. s79): the pointers of the first R planes

def vr(b, n): return f'v[{b}:{b + n - 1}]'
def mfma(d, a, b, c, op='f16'): return f'v_mfma_f32_32x32x16_{op} {vr(d, 16)}, {vr(a, 4)}, {vr(b, 4)}, ' + ('0' if c is None else vr(c, 16))

def stage_C(pi):
    o, a, b = OSET(pi), ASET(pi), BSET(pi)
    c0 = [mfma(o, WY[0, 0], b + 0, a), mfma(o, WY[0, 1], b + 4, o), mfma(o, WY[0, 0], b + 8, o), mfma(o, WY[0, 1], b + 12, o)]
    c1 = [mfma(o + 16, WY[1, 0], b + 0, a + 16), mfma(o + 16, WY[1, 1], b + 4, o + 16), mfma(o + 16, WY[1, 0], b + 8, o + 16),
          mfma(o + 16, WY[1, 1], b + 12, o + 16)]
    if SCHED == 0:
        return [c0[0], c1[0], c0[1], c1[1], c0[2], c1[2], c0[3], c1[3]]
    return c0 + c1                                     # each accumulator chain back to back (SrcC forwarding)


def stage_A():
    if DT == 'f32':          # T = (hi + lo of P) x Wx^T: the hi products first
        return [mfma(T, P[0], WX[0], None), mfma(T, P[1], WX[1], T), mfma(T, PLO[0], WX[0], T), mfma(T, PLO[1], WX[1], T)]
    if DT == 'bf16':
        return [mfma(T, P[0], WX[0], None, 'bf16'), mfma(T, P[1], WX[1], T, 'bf16'), mfma(T, PMIX, WXE, T, 'bf16')]
    return [mfma(T, P[0], WX[0], None), mfma(T, P[1], WX[1], T)]

def mix_P():
    """bf16 planes: PMIX = lanes 0..31 ? P[0] : P[1]"""
    return [f'v_cndmask_b32_e64 v{PMIX + w}, v{P[1] + w}, v{P[0] + w}, s[{S_LOW32}]' for w in range(4)]

def split_P_hi():
    """f32 planes: hi = fp16(P) -> the A operands P[0] (k-step 0: raw 0..7), P[1] (k-step 1: raw 8..15)"""
    return [f'v_cvt_pk_f16_f32 v{P[0] + w}, v{PRAW + 2 * w}, v{PRAW + 2 * w + 1}' for w in range(8)]

def split_P_lo():
    """lo = fp16(P - hi): the residual in place (exact in f32), then packed"""
    L = []
    for j in range(16):
        sel = ' op_sel:[1,0,0]' if j & 1 else ''
        L.append(f'v_fma_mix_f32 v{PRAW + j}, v{P[0] + j // 2}, -1.0, v{PRAW + j}{sel} op_sel_hi:[1,0,0]')
    return L + [f'v_cvt_pk_f16_f32 v{PLO[0] + w}, v{PRAW + 2 * w}, v{PRAW + 2 * w + 1}' for w in range(8)]

def stage_D(s):
    o, a = OSET(s), ASET(s)
    return [f'v_max_i32 v{a + r}, v{o + r}, v{a + r}' for r in range(32)]

def stage_B(s):
    b = BSET(s)
    L = [f'v_cvt_pk_f16_f32 v{b + w}, v{T + 2 * w}, v{T + 2 * w + 1}' for w in range(8)]            # hi (bhi0 = +0..3, bhi1 = +4..7)
    for j in range(16):                                                                              # lo = T - hi, in place
        sel = ' op_sel:[1,0,0]' if j & 1 else ''
        L.append(f'v_fma_mix_f32 v{T + j}, v{b + j // 2}, -1.0, v{T + j}{sel} op_sel_hi:[1,0,0]')
    L += [f'v_cvt_pk_f16_f32 v{b + 8 + w}, v{T + 2 * w}, v{T + 2 * w + 1}' for w in range(8)]        # lo
    return L

def dma(base):
    """this wave's half of a plane -> ring slot S_DMASLOT (S_RING already points at the wave's half of slot 0), advance the slot"""
    return ([f's_add_u32 m0, s{S_RING}, s{S_DMASLOT}', 's_nop 0', f'global_load_lds_dwordx4 v{GOFF}, s[{base}]{NT}'] +
            ([f'global_load_lds_dwordx4 v{GOFF}, s[{base}] offset:2048{NT}'] if LOADS == 2 else []) +
            [f's_add_u32 s{S_DMASLOT}, s{S_DMASLOT}, {SLOT}', f's_and_b32 s{S_DMASLOT}, s{S_DMASLOT}, {R * SLOT - 1}'])

def next_key():
    return ([f's_load_dwordx2 s[{S_BASE}], s[{S_KEYS}], s{S_KOFF}', f's_add_u32 s{S_KOFF}, s{S_KOFF}, 8'] +
            ([f's_memtime s[{S_TIME}]'] if FAIR else []))


def take_turns():
    """behind the lgkmcnt(0) that also returned the clock read: priority 1 while clock bit FAIR == this wave's slot parity"""
    if not FAIR:
        return []
    _label[0] += 1
    n = _label[0]
    return [f's_lshr_b32 s{S_TMP}, s{S_TIME.split(":")[0]}, {FAIR}', f's_xor_b32 s{S_TMP}, s{S_TMP}, s{S_PRIO_ID}', f's_bitcmp1_b32 s{S_TMP}, 0',
            f's_cbranch_scc1 L_hi{n}_%=', 's_setprio 0', f's_branch L_pd{n}_%=', f'L_hi{n}_%=:', 's_setprio 1', f'L_pd{n}_%=:']

def read_plane():
    """after the caller's vmcnt wait: the partner's half has landed once both waves are past the barrier"""
    if ABLATE == 3:
        return []
    if DT == 'f32':
        rd = [f'v_add_u32 v{LDS_TMP4 + q}, s{S_RDSLOT}, v{LDS_RD4 + q}' for q in range(4)]
        rd += [f'ds_read_b128 {vr(PRAW + 4 * q, 4)}, v{LDS_TMP4 + q}' for q in range(4)]
    else:
        rd = [f'v_add_u32 v{LDS_TMP}, s{S_RDSLOT}, v{LDS_RD}', f'ds_read_b128 {vr(P[0], 4)}, v{LDS_TMP}',
              f'ds_read_b128 {vr(P[1], 4)}, v{LDS_TMP} offset:32']
    return ([] if ABLATE == 2 else ['s_barrier']) + rd + [
            f's_add_u32 s{S_RDSLOT}, s{S_RDSLOT}, {SLOT}', f's_and_b32 s{S_RDSLOT}, s{S_RDSLOT}, {R * SLOT - 1}']

GAPS = {0: [6, 6, 6, 7, 6, 7, 6, 7, 6, 7], 1: [0, 0, 0, 24, 0, 0, 0, 24, 0, 16], 2: [0, 0, 0, 24, 0, 0, 0, 24, 0, 16],
        3: [0, 12, 0, 12, 0, 12, 0, 12, 0, 16]}[SCHED]       # VALU behind each MFMA (64 per iteration)

def iteration_f32(pi, do_c=True, do_d=True, do_b=True, do_a=True, do_dma=True):
    """f32 planes: 12 MFMA (8 C + 4 A) and 96 VALU per steady iteration, 8 behind every MFMA:
         gaps 0-1  the first 16 clamps of D(i-1)        (T of the previous A is not readable yet)
         gaps 2-5  B(i+1): hi / lo split of T           (complete before A(i+2) rewrites T at MFMA 8)
         gap  6    lgkmcnt(0) [plane i+2 is in PRAW], the plane DMA, hi of P -> the A operands of MFMAs 8, 9
         gaps 7-9  next key pointer; lo of P (residuals in place, then packed: the operands of MFMAs 10, 11 -- the pack of the first
                   k-step's lo is four instructions ahead of MFMA 10, the second k-step's a whole gap ahead of MFMA 11)
         gaps 10-11 the other 16 clamps"""
    L = []
    if do_a:
        L += [f's_waitcnt vmcnt({LOADS * (R - 2)})'] + read_plane()
    m = (stage_C(pi) if do_c else []) + (stage_A() if do_a else [])
    d = stage_D(1 - pi) if do_d else []
    b = stage_B(1 - pi) if do_b else []
    if do_c and do_a:
        assert do_dma
        valu = d[:16] + b + ['LGKM'] + split_P_hi() + ['KEY'] + split_P_lo() + d[16:]
        per = 8
    else:
        assert not do_a and not do_dma
        valu = d[:12] + b + d[12:]
        per = (len(valu) + len(m) - 1) // len(m)
    k = 0
    for ins in m:
        L.append(ins)
        n = 0
        while k < len(valu) and n < per:
            v = valu[k]
            k += 1
            if v == 'LGKM':
                L += ['s_waitcnt lgkmcnt(0)'] + dma(S_BASE)
            elif v == 'KEY':
                L += next_key()
            else:
                L.append(v)
                n += 1
    assert k == len(valu), (k, len(valu))
    return L


def iteration(pi, do_c=True, do_d=True, do_b=True, do_a=True, do_dma=True):
    """one pipeline step for parity pi: C(i) | D(i-1) on set 1-pi | B(i+1) -> set 1-pi | A(i+2)"""
    if DT == 'f32':
        return iteration_f32(pi, do_c, do_d, do_b, do_a, do_dma)
    L = []
    if do_a:
        L += ([] if ABLATE == 3 else [f's_waitcnt vmcnt({LOADS * (R - 2)})']) + read_plane()
    m = (stage_C(pi) if do_c else []) + (stage_A() if do_a else [])
    d = stage_D(1 - pi) if do_d else []
    b = stage_B(1 - pi) if do_b else []
    if ABLATE == 4 and do_c and do_a:
        d, b = [], []
    if ABLATE == 1 and do_c and do_a:
        do_dma = False
    # VALU order: the first 12 clamps (gaps 0-1: T of the previous A is not readable yet), the split (gaps 2-6, complete
    # before A rewrites T), the remaining clamps
    valu = d[:12] + b + d[12:]
    gaps = GAPS
    if DT == 'bf16' and do_a:
        # the four selects of PMIX open gap 6 (behind the lgkmcnt(0) at the end of gap 5); 11 MFMAs, 68 VALU
        assert do_c and SCHED == 0 and not ABLATE
        valu = valu[:38] + mix_P() + valu[38:]
        gaps = [6, 6, 6, 7, 6, 7, 8, 8, 5, 5, 4]
    k = 0
    for gi, ins in enumerate(m):
        steady = do_c and do_a
        if SCHED == 2 and steady and gi in (0, 4, 8):
            L.append('s_setprio 1')
        if not (ABLATE == 5 and steady):
            L.append(ins)
        if SCHED == 2 and steady and gi in (3, 7, 9):
            L.append('s_setprio 0')
        n = gaps[gi] if (do_c and do_a) else (len(valu) + len(m) - 1) // len(m)
        L += valu[k:k + n]
        k += n
        # the plane DMA (behind the lgkmcnt(0) that retires this iteration's plane reads and last iteration's key load) and
        # the next key load: inside VALU blocks, never between the MFMAs of a chain
        if do_dma and do_c and gi == (5 if SCHED == 0 else 3):
            L += ['s_waitcnt lgkmcnt(0)'] + take_turns() + dma(S_BASE)
        if do_dma and do_c and gi == (6 if SCHED == 0 else 7):
            L += next_key()
        if SCHED != 0 and ABLATE == 1 and do_c and do_a and gi == 3:
            L += ['s_waitcnt lgkmcnt(0)']
    L += valu[k:]
    return L

def build_prefill():
    """statement 1: the ring's first R planes (this wave's halves) are on their way; nothing is waited for"""
    L = [f's_mov_b32 s{S_M0SAVE}, m0', f's_load_dwordx16 s[{S_PRE}:{S_PRE + 15}], s[{S_KEYS}], 0x0']
    if R == 16:
        L += [f's_load_dwordx16 s[{S_PRE + 16}:{S_PRE + 31}], s[{S_KEYS}], 0x40']
    L += ['s_waitcnt lgkmcnt(0)']
    for q in range(R):
        L += [f's_add_u32 m0, s{S_RING}, 0x{q * SLOT:x}', 's_nop 0', f'global_load_lds_dwordx4 v{GOFF}, s[{S_PRE + 2 * q}:{S_PRE + 2 * q + 1}]{NT}']
        if LOADS == 2:
            L += [f'global_load_lds_dwordx4 v{GOFF}, s[{S_PRE + 2 * q}:{S_PRE + 2 * q + 1}] offset:2048{NT}']
    L += [f's_mov_b32 m0, s{S_M0SAVE}']
    return L


def build():
    L = [f's_mov_b32 s{S_M0SAVE}, m0'] + ([f's_mov_b32 s{S_LOW32.split(":")[0]}, -1', f's_mov_b32 s{S_LOW32.split(":")[1]}, 0'] if DT == 'bf16' else [])
    # the odd planes' running sums and the o set the first D reads: zero (the even planes' sums come in: the same-size keys)
    L += [f'v_mov_b32 v{r}, 0' for r in list(range(ASET(1), ASET(1) + 32)) + list(range(OSET(1), OSET(1) + 32))]
    L += [f's_load_dwordx2 s[{S_BASE}], s[{S_KEYS}], 0x{8 * R:x}',
          f's_mov_b32 s{S_KOFF}, 0x{8 * (R + 1):x}', f's_mov_b32 s{S_RDSLOT}, 0', f's_mov_b32 s{S_DMASLOT}, 0']
    split = (split_P_hi() + split_P_lo() + ['s_nop 1']) if DT == 'f32' else (mix_P() + ['s_nop 1']) if DT == 'bf16' else []   # f32 planes: the A operands are made, not read
    # i = -2: A(0)
    L += [f's_waitcnt vmcnt({LOADS * (R - 1)})'] + read_plane() + ['s_waitcnt lgkmcnt(0)'] + split + stage_A()
    # i = -1: B(0) -> set 0, A(1); DMA of plane R into slot 0
    L += [f's_waitcnt vmcnt({LOADS * (R - 2)})'] + read_plane()
    L += ['s_nop 15', 's_nop 15'] + (['s_nop 15', 's_nop 15'] if DT == 'f32' else [])     # T(0): the MFMAs above -> first VALU read
    L += stage_B(0)
    L += ['s_waitcnt lgkmcnt(0)'] + split + stage_A() + dma(S_BASE) + next_key()
    # steady state: i = 0 .. NK-3, two iterations per trip
    L += ['L_fin_top%=:'] + iteration(0) + iteration(1)
    L += [f's_sub_u32 s{S_TRIPS}, s{S_TRIPS}, 1', f's_cmp_lg_u32 s{S_TRIPS}, 0', 's_cbranch_scc1 L_fin_top%=']
    # i = NK-2 (parity 0): C, D(NK-3), B(NK-1); i = NK-1 (parity 1): C, D(NK-2); then D(NK-1)
    L += iteration(0, do_a=False, do_dma=False)
    L += iteration(1, do_a=False, do_b=False, do_dma=False)
    L += ['s_nop 15', 's_nop 15'] + stage_D(1)
    L += ['s_waitcnt vmcnt(0)', f's_mov_b32 m0, s{S_M0SAVE}'] + (['s_setprio 0'] if FAIR else [])
    return L


def emit(path, header, lines, outs, ins, clob):
    out = header + ['asm volatile(']
    for l in lines:
        out.append(f'    "{l}\\n\\t"')
    out.append('    : ' + outs)
    out.append('    : ' + ins)
    out.append('    : ' + ', '.join(clob) + ');')
    open(path, 'w').write('\n'.join(out) + '\n')


def configure(dt):
    global DT, R, SLOT, LOADS
    DT = dt
    R, SLOT, LOADS = (8, 4096, 2) if dt == 'f32' else (int(os.environ.get('DAAM_PIPE_RING', '16')), 2048, 1)
    assert R in (8, 16)


def write(dt, here):
    configure(dt)
    lab = bool(ABLATE or os.environ.get('DAAM_PIPE_OUT'))
    tag = f'r{R}' if dt == 'f16' else dt            # the fp16 files keep their names (r16); bf16 shares the fp16 prefill
    pre = build_prefill()
    if not ABLATE and dt != 'bf16':
        emit(os.path.join(here, f'daam_finalize_pipe_prefill_{tag}.inc' if not os.environ.get('DAAM_PIPE_OUT') else 'daam_finalize_pipe_prefill_ablx.inc'),
             ['// GENERATED by tools/gen_fin_pipe.py -- do not edit.  Statement 1: LDS-DMA of the first planes of the ring (this wave\'s halves).'],
             pre, '', '"{v208}"(goff), "{s[36:37]}"(key_ptrs), "{s42}"(ring_half)',
             [f'"s{r}"' for r in [S_M0SAVE] + list(range(S_PRE, S_PRE + 2 * R))] + ['"memory"', '"scc"'])
    lines = build()
    used_v = sorted(set(range(P[0], GOFF)) - set(range(ASET(0), ASET(0) + 64)))
    used_v += (list(range(PLO[0], PRAW + 16)) + list(range(LDS_TMP4, LDS_TMP4 + 4))) if dt == 'f32' else [LDS_TMP] + (list(range(PMIX, PMIX + 4)) if dt == 'bf16' else [])
    clob = [f'"v{r}"' for r in used_v] + [f'"s{r}"' for r in [S_KOFF, 40, 41, S_RDSLOT, S_DMASLOT, 45, S_M0SAVE] + ([50, 51, S_TMP] if FAIR else []) + ([52, 53] if dt == 'bf16' else [])]
    clob += ['"memory"', '"scc"', '"vcc"']
    rd = f'"{{v[{LDS_RD4}:{LDS_RD4 + 3}]}}"(lds_rd4)' if dt == 'f32' else '"{v210}"(lds_rd)'
    emit(os.path.join(here, f'daam_finalize_pipe_asm_{tag}.inc' if not lab else (os.environ.get('DAAM_PIPE_OUT') or f'daam_finalize_pipe_asm_abl{ABLATE}.inc')),
         ['// GENERATED by tools/gen_fin_pipe.py -- do not edit; the schedule and its hazard distances are documented there.',
          '// Statement 2: software-pipelined loop over the planes of this workgroup\'s chunk (the ring was started by statement 1), drain.',
          f'// {sum(1 for l in lines if l.startswith("v_mfma"))} MFMA + {sum(1 for l in lines if l.startswith("v_") and not l.startswith("v_mfma"))} VALU statements in the text; ring of {R} planes per workgroup.'
          + ('' if dt == 'f16' else f'  Planes: {dt}.')],
         lines,
         '"+{v[144:159]}"(accA0), "+{v[160:175]}"(accA1), "={v[176:191]}"(accB0), "={v[192:207]}"(accB1), "+{s39}"(trips)',
         '"{v[0:3]}"(wx0), "{v[4:7]}"(wx1), "{v[8:11]}"(wy00), "{v[12:15]}"(wy01), "{v[16:19]}"(wy10), "{v[20:23]}"(wy11),\n'
         f'      "{{v208}}"(goff), {rd}, "{{s[36:37]}}"(key_ptrs), "{{s42}}"(ring_half)' + (', "{s46}"(prio_id)' if FAIR else '')
         + (f', "{{v[{WXE}:{WXE + 3}]}}"(wxe)' if dt == 'bf16' else ''), clob)
    per_iter = iteration(0)
    print(dt, 'wrote', here, len(pre), '+', len(lines), 'instructions;', 'steady iteration:', len(per_iter), 'instructions,',
          sum(1 for l in per_iter if l.startswith('v_mfma')), 'MFMA')


def main():
    here = os.environ.get('DAAM_PIPE_OUTDIR') or os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'daam_amd', 'csrc')
    for dt in ([os.environ['DAAM_PIPE_DT']] if os.environ.get('DAAM_PIPE_DT') else ['f16', 'bf16', 'f32']):
        write(dt, here)


if __name__ == '__main__':
    main()
