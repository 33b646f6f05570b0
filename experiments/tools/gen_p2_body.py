#!/usr/bin/env python3
"""Generates experiments/csrc/gemm_p2_body{1,2}.inc: the fully unrolled work-item body of the one-wave-per-SIMD GEMM whose epilogue runs in
the MFMA gaps of the NEXT item (gemm_p2.hip, variant 14).  The schedule -- which fragment read, LDS-DMA piece, epilogue step, store
and accumulator copy follows which MFMA -- and every counted s_waitcnt vmcnt(N) are decided HERE, in one place, and the counts are
derived from the generated instruction order itself (simulate()), not by hand.

    python tools/gen_p2_body.py            # rewrites the .inc (committed; the build does not run this script)

Tile 256 x 128, 4 waves of 128 x 64 (2 x 2), k-tile 64, NT = 12 k-tiles (K = 768), v_mfma_f32_16x16x32_bf16.
Per k-tile 4 phases of 16 MFMAs: (A0,B0) (A0,B1) (A1,B1) (A1,B0); MFMA j of a phase: ks = j >> 3, mb = (j >> 1) & 3, nbl = j & 1.
Phase p reads the fragment set phase p + 1 starts with (gaps 0..7), issues the LDS-DMA pieces of k-tile kt + 2 (gaps 8..11:
A0 4 pieces, B0 2, B1 2, A1 4) and, in gaps 12..15, the epilogue's global stores.  Barrier X at phase 0 (first half F = A0, B0 of
k-tile kt consumed, second half S = B1, A1 landed), barrier Y at phase 2 (S consumed, F of kt + 1 landed).
Epilogue of the PREVIOUS item (accumulators copied to VGPRs P during the last k-tile): 32 pieces (one 16 x 16 accumulator block =
4 values per lane) x 18 steps in the gaps of MFMAs 32 .. 607; a 32-row block of 8 pieces leaves through the wave's LDS transposition
buffers (4 reads per stream in the two gaps after its last piece) and is stored one phase later.
"""
import os

NT = 12
PH = 4 * NT                      # phases per item
STEPS = 18                       # epilogue steps per piece
EP0 = 16                         # first MFMA gap that carries an epilogue step (behind the last copies into P; the bias landed before barrier X of k-tile 0)
BIAS_PHASE = 42                  # the bias loads follow the B1 pieces of the next item's k-tile 0: older than its A1 pieces, so barrier X_0's wait covers them
DMA_PIECES = {0: ("A0", 4), 1: ("B0", 2), 2: ("B1", 2), 3: ("A1", 4)}     # phase -> unit staged, pieces per wave
UNIT_OFF = {"A0": 0, "B0": 16384, "B1": 24576, "A1": 32768}
BUF = 49152
PHASE_AB = {0: (0, 0), 1: (0, 1), 2: (1, 1), 3: (1, 0)}                   # phase -> (A half, B half)


def block_of_piece(pz):
    b, k = divmod(pz, 8)
    hh, i = divmod(b, 2)
    mbl, nb = divmod(k, 4)
    return b, hh, 2 * i + mbl, nb, mbl


def tpr_gap(b):                  # MFMA index after which block b's transposition reads are issued (two consecutive gaps)
    return EP0 + STEPS * 8 * (b + 1)


def store_phase(b):
    """first phase behind the transposition reads that stages a B unit (two LDS-DMA pieces per wave, gaps 8 and 9): its gaps
    12..15 carry the stores"""
    ph = tpr_gap(b) // 16 + 1
    while ph % 4 not in (1, 2):
        ph += 1
    assert ph < 40
    return ph


def build(two_streams):
    """-> list of (phase, gap, kind, text) in program order; kind in {'mf','rd','dma','st','bias','ep','cp','tpr','setup'}"""
    ops = []
    for kt in range(NT):
        par = kt & 1
        b0set, b1set = ("bx", "by") if par == 0 else ("by", "bx")
        for p in range(4):
            ph = 4 * kt + p
            ah, bh = PHASE_AB[p]
            aset = "a0" if ah == 0 else "a1"
            bset = b0set if bh == 0 else b1set
            # ---- what this phase prefetches (for the next phase)
            if p == 0:
                reads = [("B", b1set, blk, ks, par, UNIT_OFF["B1"]) for ks in range(2) for blk in range(2)]
            elif p == 1:
                reads = [("A", "a1", blk, ks, par, UNIT_OFF["A1"]) for ks in range(2) for blk in range(4)]
            elif p == 2:
                reads = [("A", "a0", blk, ks, par ^ 1, UNIT_OFF["A0"]) for ks in range(2) for blk in range(4)]
            else:
                reads = [("B", b1set, blk, ks, par ^ 1, UNIT_OFF["B0"]) for ks in range(2) for blk in range(2)]
            unit, npieces = DMA_PIECES[p]
            kts = (kt + 2) % NT
            cross = 1 if kt + 2 >= NT else 0
            for j in range(16):
                m = 16 * ph + j
                ks, mb, nbl = j >> 3, (j >> 1) & 3, j & 1
                nb = 2 * bh + nbl
                mac = "P2_MF0" if (kt == 0 and ks == 0) else "P2_MF"
                ops.append((ph, j, "mf", "%s(%d, %d, %d, %s[%d][%d], %s[%d][%d]);" % (mac, ah, mb, nb, bset, nbl, ks, aset, mb, ks)))
                if j < len(reads):
                    kind, dst, blk, ks_r, buf, uoff = reads[j]
                    ops.append((ph, j, "rd", "P2_RD%s(%s[%d][%d], %d, %d, %d);" % (kind, dst, blk, ks_r, ks_r, buf, blk * 2048 + uoff)))
                # LDS-DMA pieces of this phase's unit in gaps 0 .. npieces-1 of the SECOND half (behind the fragment reads).
                # (Tried: staggered -- in gap j only wave j % 4 issues, one request per MFMA interval on the CU's address path
                #  instead of four at once: the 768 wave-uniform branches per item cost more than the contention they avoid,
                #  plain main loop 395 -> 542 us at 98304 x 3072 x 768; profiles/r05_experiments.md)
                if 8 <= j < 8 + npieces:
                    ops.append((ph, j, "dma", "P2_DMA_%s(%d, %d, %d, 0);" % (unit, j - 8, par, kts)))
                if kt == NT - 3 and p == 3 and j == 15:
                    ops.append((ph, j, "setup", "P2_SETUP_NEXT();"))
                # ---- epilogue of the pending item
                if EP0 <= m < EP0 + 32 * STEPS:
                    pz, s = divmod(m - EP0, STEPS)
                    ops.append((ph, j, "ep", "P2_EP(%d, %d);" % (pz, s)))
                for b in range(4):
                    if m == tpr_gap(b):
                        ops.append((ph, j, "tpr", "P2_TPR(%d, 0);" % b))
                    if m == tpr_gap(b) + 1 and two_streams:
                        ops.append((ph, j, "tpr", "P2_TPR(%d, 1);" % b))
                    if ph == store_phase(b) and j >= 12:
                        # behind the phase's two LDS-DMA pieces: gaps 12..15, output stream then second stream
                        ops.append((ph, j, "st", "P2_ST(%d, %d, 0);" % (b, j - 12)))
                        if two_streams:
                            ops.append((ph, j, "st", "P2_ST(%d, %d, 1);" % (b, j - 12)))
                if ph == BIAS_PHASE and 12 <= j < 16:
                    ops.append((ph, j, "bias", "P2_BIASLD(%d);" % (j - 12)))
                # ---- accumulators of the item being finished -> P, two values per gap, one phase behind their last MFMA
                cp = None
                if kt == NT - 1 and p >= 1:
                    cp = PHASE_AB[p - 1]
                elif kt == 0 and p == 0:
                    cp = PHASE_AB[3]
                if cp is not None:
                    cah, cbh = cp
                    blk = j // 2
                    cmb, cnbl = blk >> 1, blk & 1
                    ops.append((ph, j, "cp", "P2_CP(%d, %d, %d, %d);" % (cah, cmb, 2 * cbh + cnbl, 2 * (j & 1))))
    return ops


def simulate(ops):
    """vmcnt(N) of the two barriers of every k-tile, with and without the epilogue's stores (first item of a workgroup):
    N = vector-memory operations issued after the LAST piece of the half that must have landed, up to the barrier.
    The stream is periodic (the prologue of gemm_p2.hip issues phases 40..47's operations in their order)."""
    def events(with_stores):
        ev = []
        for (ph, j, kind, text) in ops:
            if kind == "dma":
                ev.append((ph, j, "dma", text.split("(")[0].split("_")[-1]))
            elif kind == "bias" or (kind == "st" and with_stores):
                ev.append((ph, j, "other", None))
        return ev
    out = {}
    for kt in range(NT):
        for which in ("X", "Y"):
            row = []
            for with_stores in (True, False):
                # two periods of the stream: the previous item's tail never carries stores (all store phases < 40)
                prev = [(ph - PH, j, k, u) for (ph, j, k, u) in events(True) if ph >= 40]
                cur = events(with_stores)
                seq = prev + cur
                if which == "X":
                    wait_ph, need = 4 * kt, [(4 * (kt - 2) + 2, "B1"), (4 * (kt - 2) + 3, "A1")]
                else:
                    wait_ph, need = 4 * kt + 2, [(4 * (kt - 1), "A0"), (4 * (kt - 1) + 1, "B0")]
                last = max(i for i, (ph, j, k, u) in enumerate(seq) if k == "dma" and (ph, u) in need)
                n = sum(1 for i, (ph, j, k, u) in enumerate(seq) if i > last and ph < wait_ph)
                row.append(n)
            out[(kt, which)] = tuple(row)
    return out


def emit(two_streams, path_tag):
    ops = build(two_streams)
    waits = simulate(ops)
    lines = ["// GENERATED by tools/gen_p2_body.py (%s) -- do not edit; schedule and vmcnt counts are derived there" % path_tag]
    cur = (-1, -1)
    for (ph, j, kind, text) in ops:
        if kind == "mf":
            if cur != (-1, -1):
                lines.append("P2_GAP();")
            if j == 0:
                kt, p = divmod(ph, 4)
                lines.append("// ---- k-tile %d phase %d" % (kt, p))
                lines.append("P2_PHASE_BEGIN();")
                if p in (0, 2):
                    wp, wn = waits[(kt, "X" if p == 0 else "Y")]
                    assert wp <= 63 and wn <= 63
                    lines.append("P2_WAITB(%d, %d);" % (wp, wn))
                    if kt == 0 and p == 0:
                        lines.append("P2_BIASFIX();")
            cur = (ph, j)
        lines.append(text)
    lines.append("P2_GAP();")
    return "\n".join(lines) + "\n", waits


if __name__ == "__main__":
    root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    for two, name in ((False, "gemm_p2_body1.inc"), (True, "gemm_p2_body2.inc")):
        text, waits = emit(two, "one output stream" if not two else "two output streams")
        with open(os.path.join(root, "experiments", "csrc", name), "w") as f:
            f.write(text)
        print(name, len(text.splitlines()), "lines; waits (with stores, without):",
              " ".join("%s%d=%d/%d" % (w, kt, a, b) for (kt, w), (a, b) in sorted(waits.items())))
