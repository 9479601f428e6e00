"""Generates csrc/screen_kloop.inc: the hand-scheduled K loop of the screening kernel's two-block tiles
(xcorr_screen.hip, `step == 32`: 5..8 partners) as ONE inline-asm block with fixed registers.

Why by hand: written in C++ the loop either copies 24 fragment registers per K step (the matrix pipe then waits
for the vector port: an MFMA holds it for 8 of its 16 cycles, a v_mov for 4) or, unrolled so that nothing is
copied, exceeds the 128 VGPRs of two workgroups per CU and spills.  The formulation below needs 48 fragment
registers: tile t+2 at K step n and tile t at K step n+1 read the SAME A fragment (their byte offsets differ by
64 = one K step), so the loop walks the A stream once — fragment pair F_k = (hi, lo) at byte offset 32k — and
multiplies every pair by TWO partner fragments: F_{2n+2} by B[n] into tile 2 and by B[n+1] into tile 0, F_{2n+3}
likewise into tiles 3 and 1.  Per K step: 4 A fragments + one B pair from LDS, 12 products, no register copies;
every LDS read is requested six products (>= 96 matrix-pipe cycles) before its first use.

    python tools/gen_screen_kloop.py > narrow_band_least_squares_amd/csrc/screen_kloop.inc
"""
ACC = {'h0': 96, 'm0': 100, 'h1': 104, 'm1': 108, 'h2': 112, 'm2': 116, 'h3': 120, 'm3': 124}
FR = {'eh': 64, 'el': 68, 'oh': 72, 'ol': 76}
BB = {'b0h': 80, 'b0l': 84, 'b1h': 88, 'b1l': 92}
PAH, PAL, PBH, PBL = '%[pah]', '%[pal]', '%[pbh]', '%[pbl]'   # in/out operands (advanced by the loop)
FIRST_FIXED = 64

out = []


def emit(s):
    out.append(s)


def r4(b):
    return 'v[%d:%d]' % (b, b + 3)


def reg(name):
    for d in (ACC, FR, BB):
        if name in d:
            return r4(d[name])
    raise KeyError(name)


def mfma(acc, a, b, zero=False):
    emit('v_mfma_i32_16x16x64_i8 %s, %s, %s, %s' % (reg(acc), reg(a), reg(b), '0' if zero else reg(acc)))


def load_frag(which, off):          # which = 'e' or 'o': hi and lo limb fragments, two aligned 8-byte reads each
    for limb, ptr in (('h', PAH), ('l', PAL)):
        b = FR[which + limb]
        emit('ds_read_b64 v[%d:%d], %s offset:%d' % (b, b + 1, ptr, off))
        emit('ds_read_b64 v[%d:%d], %s offset:%d' % (b + 2, b + 3, ptr, off + 8))


def load_b(which, off):             # which = 'b0' or 'b1'
    emit('ds_read_b128 %s, %s offset:%d' % (r4(BB[which + 'h']), PBH, off))
    emit('ds_read_b128 %s, %s offset:%d' % (r4(BB[which + 'l']), PBL, off))


def wait(n):
    emit('s_waitcnt lgkmcnt(%d)' % n)


def step(c, x, oa, ob):
    """K step n with a successor.  E = F_{2n+2} (requested), c = B[n] (refilled with B[n+2]), x = B[n+1].
    LDS requests in flight at entry, oldest first: E (4), the refill of x (2)."""
    load_frag('o', 96 + oa)                       # F_{2n+3}
    wait(6)                                       # E landed
    mfma('h2', 'eh', c + 'h')
    mfma('m2', 'eh', c + 'l')
    wait(4)                                       # x landed
    mfma('h0', 'eh', x + 'h')
    mfma('m0', 'eh', x + 'l')
    mfma('m2', 'el', c + 'h')
    mfma('m0', 'el', x + 'h')
    load_frag('e', 128 + oa)                      # F_{2n+4}
    wait(4)                                       # O landed
    mfma('h3', 'oh', c + 'h')
    mfma('m3', 'oh', c + 'l')
    mfma('m1', 'oh', x + 'l')
    mfma('m3', 'ol', c + 'h')
    load_b(c, 128 + ob)                           # B[n+2] into the registers of B[n]
    mfma('h1', 'oh', x + 'h')
    mfma('m1', 'ol', x + 'h')


def tail(c, oa):
    """Last K step: tiles 2 and 3 only."""
    load_frag('o', 96 + oa)
    wait(4)
    mfma('h2', 'eh', c + 'h')
    mfma('m2', 'eh', c + 'l')
    wait(0)
    mfma('m3', 'oh', c + 'l')
    mfma('m2', 'el', c + 'h')
    mfma('h3', 'oh', c + 'h')
    mfma('m3', 'ol', c + 'h')


def two_block():
    """step == 32 (5..8 partners, two lag blocks per tile step): NBLS_SCREEN_KLOOP_ASM."""
    del out[:]
    emit('s_sub_u32 %[cnt], %[nst], 1')
    load_b('b0', 0)
    load_frag('e', 0)                             # F_0
    load_frag('o', 32)                            # F_1
    for a in ('h2', 'm2', 'h3', 'm3'):            # tiles 2,3 start from zero (tiles 0,1: literal 0 in their first product)
        for i in range(4):
            emit('v_mov_b32 v%d, 0' % (ACC[a] + i))
    wait(4)                                       # B[0], F_0 landed
    mfma('h0', 'eh', 'b0h', zero=True)
    mfma('m0', 'eh', 'b0l', zero=True)
    wait(0)                                       # F_1 landed
    mfma('m1', 'oh', 'b0l', zero=True)
    mfma('m0', 'el', 'b0h')
    load_frag('e', 64)                            # F_2
    load_b('b1', 64)                              # B[1]  (request order E, B as inside the loop)
    mfma('h1', 'oh', 'b0h', zero=True)
    mfma('m1', 'ol', 'b0h')
    emit('s_cmp_eq_u32 %[cnt], 0')
    emit('s_cbranch_scc1 2f')
    emit('1:')
    step('b0', 'b1', 0, 0)
    emit('s_sub_u32 %[cnt], %[cnt], 1')
    emit('s_cmp_eq_u32 %[cnt], 0')
    emit('s_cbranch_scc1 3f')
    step('b1', 'b0', 64, 64)
    for p in (PAH, PAL, PBH, PBL):
        emit('v_add_u32 %s, 0x80, %s' % (p, p))
    emit('s_sub_u32 %[cnt], %[cnt], 1')
    emit('s_cmp_lg_u32 %[cnt], 0')
    emit('s_cbranch_scc1 1b')
    emit('2:')
    tail('b0', 0)
    emit('s_branch 4f')
    emit('3:')
    tail('b1', 64)
    emit('4:')
    emit('s_nop 15')                              # matrix-core results -> vector reads: no hardware interlock
    emit('s_nop 7')
    clob = ', '.join('"v%d"' % i for i in range(FIRST_FIXED, 96))
    print('// GENERATED by tools/gen_screen_kloop.py — do not edit.  Fixed registers: v64-79 A fragment')
    print('// pairs (E, O), v80-95 partner fragments (b0, b1), v96-127 the eight accumulators (outputs).')
    print('#define NBLS_SCREEN_KLOOP_ASM(H0, M0, H1, M1, H2, M2, H3, M3, PAH, PAL, PBH, PBL, NST, CNT) \\')
    print('    asm volatile( \\')
    for ln in out:
        print('        "%s\\n\\t" \\' % ln)
    outs = ', '.join('"=&{v[%d:%d]}"(%s)' % (ACC[k], ACC[k] + 3, k.upper()) for k in ('h0', 'm0', 'h1', 'm1', 'h2', 'm2', 'h3', 'm3'))
    print('        : %s, [cnt] "=&s"(CNT), [pah] "+v"(PAH), [pal] "+v"(PAL), [pbh] "+v"(PBH), [pbl] "+v"(PBL) \\' % outs)
    print('        : [nst] "s"(NST) \\')
    print('        : %s, "scc", "memory")' % clob)


# ------------------------------------------------------------------------------------------------------------
# One lag block per tile step (step == 16: 9..17 partners per group), EIGHT tiles per group, for the workgroups that
# have a CU to themselves (two waves per SIMD: 256 VGPRs).  Tile t at K step n reads the A fragment pair
# F[4n + t] (byte offset 16 per index), so F[4n+4 .. 4n+7] serve tiles 4..7 at step n AND tiles 0..3 at step n+1:
# per K step 4 fragment pairs + one partner pair from LDS for 24 products — half the LDS bytes per product of the
# four-tile loop, which ran at 83 % of the LDS bandwidth.  Ring of four fragment pairs, three requests in flight;
# the wait counts come from a simulation of the (in-order) LDS return queue.
class S1:
    H = [192 + 8 * t for t in range(8)]           # accumulators of tile t: H at 192 + 8t, M at 196 + 8t
    M = [196 + 8 * t for t in range(8)]
    R = [(160 + 8 * u, 164 + 8 * u) for u in range(4)]      # ring of fragment pairs (hi, lo)
    B = {'b0': (144, 148), 'b1': (152, 156)}

    def __init__(self):
        self.q = []                               # outstanding LDS requests, oldest first: names

    def load_pair(self, u, k, base):              # R[u] <- F[k]; offsets relative to F[base]
        off = 16 * (k - base)
        for b, ptr in ((self.R[u][0], PAH), (self.R[u][1], PAL)):
            emit('ds_read_b64 v[%d:%d], %s offset:%d' % (b, b + 1, ptr, off))
            emit('ds_read_b64 v[%d:%d], %s offset:%d' % (b + 2, b + 3, ptr, off + 8))
            self.q += ['R%d' % u, 'R%d' % u]
        assert len(self.q) <= 15, self.q

    def load_b(self, name, idx, base):
        off = 64 * (idx - base)
        emit('ds_read_b128 %s, %s offset:%d' % (r4(self.B[name][0]), PBH, off))
        emit('ds_read_b128 %s, %s offset:%d' % (r4(self.B[name][1]), PBL, off))
        self.q += [name, name]
        assert len(self.q) <= 15, self.q

    def need(self, *names):                       # wait until every request that writes one of `names` has returned
        last = max([i for i, n in enumerate(self.q) if n in names], default=-1)
        if last >= 0:
            emit('s_waitcnt lgkmcnt(%d)' % (len(self.q) - 1 - last))
            self.q = self.q[last + 1:]

    def prod(self, acc, a, b, zero=False):        # one limb product: acc += a x b (registers by base index)
        emit('v_mfma_i32_16x16x64_i8 %s, %s, %s, %s' % (r4(acc), r4(a), r4(b), '0' if zero else r4(acc)))

    def mf(self, u, b, t, zero=False):            # the three limb products of ONE tile: fragment pair R[u] x partner pair b
        (rh, rl), (bh, bl) = self.R[u], self.B[b]     # (the two products into M are kept apart: no back-to-back dependence)
        self.prod(self.M[t], rh, bl, zero)
        self.prod(self.H[t], rh, bh, zero)
        self.prod(self.M[t], rl, bh)

    def step(self, bc, bx, n, base):              # K step n (relative numbering: F[base], B[base/4] at offset 0)
        for u in range(4):
            (rh, rl), (ch, cl), (xh, xl) = self.R[u], self.B[bc], self.B[bx]
            ta, tb = 4 + u, u                     # tile 4+u with B[n], tile u with B[n+1]
            self.need('R%d' % u)
            self.prod(self.H[ta], rh, ch)
            self.prod(self.M[ta], rh, cl)
            if u == 0:
                self.need(bx)
            if u < 3:
                self.prod(self.H[tb], rh, xh)
                self.prod(self.M[tb], rh, xl)
                self.prod(self.M[ta], rl, ch)
                self.prod(self.M[tb], rl, xh)
            else:                                 # last use of B[n]: its registers are refilled with B[n+2] at once
                self.prod(self.M[tb], rh, xl)
                self.prod(self.M[ta], rl, ch)
                self.load_b(bc, n + 2, base // 4)
                self.prod(self.H[tb], rh, xh)
                self.prod(self.M[tb], rl, xh)
            self.load_pair((u + 3) & 3, 4 * n + u + 7, base)

    def tail(self, bc, n, base):
        for u in range(4):
            self.need('R%d' % u)
            self.mf(u, bc, 4 + u)
            if u == 0:
                self.load_pair(3, 4 * n + 7, base)


def one_block():
    """step == 16, eight tiles per group: NBLS_SCREEN_KLOOP_S1_ASM (kernel instance with 256 VGPRs)."""
    del out[:]
    g = S1()
    emit('s_sub_u32 %[cnt], %[nst], 1')
    g.load_b('b0', 0, 0)
    for u in range(3):
        g.load_pair(u, u, 0)                      # F[0..2]
    for t in range(4, 8):                         # tiles 4..7 start from zero (tiles 0..3: literal 0 in their first product)
        for i in range(8):
            emit('v_mov_b32 v%d, 0' % (S1.H[t] + i))
    g.need('R0', 'b0')
    g.mf(0, 'b0', 0, zero=True)
    g.load_pair(3, 3, 0)                          # F[3]
    g.need('R1')
    g.mf(1, 'b0', 1, zero=True)
    g.load_pair(0, 4, 0)                          # slot 0 of K step 0
    g.need('R2')
    g.mf(2, 'b0', 2, zero=True)
    g.load_pair(1, 5, 0)
    g.need('R3')
    g.mf(3, 'b0', 3, zero=True)
    g.load_b('b1', 1, 0)                          # (request order R0, R1, B, R2 as at the top of every K step)
    g.load_pair(2, 6, 0)
    top = list(g.q)
    emit('s_cmp_eq_u32 %[cnt], 0')
    emit('s_cbranch_scc1 2f')
    emit('1:')
    g.step('b0', 'b1', 0, 0)
    mid = list(g.q)
    emit('s_sub_u32 %[cnt], %[cnt], 1')
    emit('s_cmp_eq_u32 %[cnt], 0')
    emit('s_cbranch_scc1 3f')
    g.step('b1', 'b0', 1, 0)
    for p in (PAH, PAL, PBH, PBL):
        emit('v_add_u32 %s, 0x80, %s' % (p, p))
    emit('s_sub_u32 %[cnt], %[cnt], 1')
    emit('s_cmp_lg_u32 %[cnt], 0')
    emit('s_cbranch_scc1 1b')
    swap = {'b0': 'b1', 'b1': 'b0'}
    assert g.q == top and [swap.get(x, x) for x in mid] == top, (top, mid, g.q)     # the queue state is periodic
    emit('2:')
    g.q = list(top)
    g.tail('b0', 0, 0)
    g.need('R3')
    emit('s_branch 4f')
    emit('3:')
    g.q = list(mid)
    g.tail('b1', 1, 0)
    g.need('R3')
    assert not [x for x in g.q if x.startswith('R')]
    emit('4:')
    emit('s_waitcnt lgkmcnt(0)')                  # (a partner refill of the last whole step may still be on its way)
    emit('s_nop 15')
    emit('s_nop 7')
    clob = ', '.join('"v%d"' % i for i in range(144, 192))
    print()
    print('// One lag block per tile step, eight tiles per group (kernel instance with 256 VGPRs).  Fixed registers: v144-159')
    print('// partner fragments (b0, b1), v160-191 ring of four A fragment pairs, v192-255 the sixteen accumulators (outputs).')
    names = ', '.join('H%d, M%d' % (t, t) for t in range(8))
    print('#define NBLS_SCREEN_KLOOP_S1_ASM(%s, PAH, PAL, PBH, PBL, NST, CNT) \\' % names)
    print('    asm volatile( \\')
    for ln in out:
        print('        "%s\\n\\t" \\' % ln)
    outs = ', '.join('"=&{v[%d:%d]}"(H%d), "=&{v[%d:%d]}"(M%d)' % (S1.H[t], S1.H[t] + 3, t, S1.M[t], S1.M[t] + 3, t) for t in range(8))
    print('        : %s, [cnt] "=&s"(CNT), [pah] "+v"(PAH), [pal] "+v"(PAL), [pbh] "+v"(PBH), [pbl] "+v"(PBL) \\' % outs)
    print('        : [nst] "s"(NST) \\')
    print('        : %s, "scc", "memory")' % clob)


two_block()
one_block()
