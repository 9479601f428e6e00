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


def main():
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


main()
