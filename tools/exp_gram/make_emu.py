"""Experiment: does an expansion-form / Gram-table evaluation of the search's
dot products keep the codes the reference returns?  Builds a patched copy of oracle/mcq_oracle.c in
/tmp whose pair-stage dots (and optionally the stage-0 cross term) are evaluated the way a Gram-table
implementation would, and counts mismatches against every reference fixture.

  GRAM_PAIR = 0 oracle as is | 1 fp64-accurate dots | 2 fp32 (correctly rounded) Gram entries combined in fp32 |
              3 fp32 MFMA-chain Gram entries combined in fp32
  GRAM_S0   = 0 oracle as is | 1 X = fl32(fp64 Gram sums - fp32 x.c chain) | 2 rounded entries | 3 fp32 chain entries, fp32 sums
"""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
src = open(os.path.join(ROOT, "oracle", "mcq_oracle.c")).read()

# 1. pair dots
old_pair = """            for (int p = 0; p < M; p++) acc[p] = 0.0f;
            for (int i = 0; i < Dp; i++) {
                const float *ea = de + (size_t)i * Kg, *ob = dod + (size_t)i * Kg;
                for (int a = 0; a < Kg; a++) {
                    float av = ea[a];
                    float *row = acc + (size_t)a * Kg;
                    for (int b = 0; b < Kg; b++) row[b] = fmaf(av, ob[b], row[b]);
                }
            }
"""
new_pair = """            for (int p = 0; p < M; p++) acc[p] = 0.0f;
            if (g_pair_mode == 0) {
            for (int i = 0; i < Dp; i++) {
                const float *ea = de + (size_t)i * Kg, *ob = dod + (size_t)i * Kg;
                for (int a = 0; a < Kg; a++) {
                    float av = ea[a];
                    float *row = acc + (size_t)a * Kg;
                    for (int b = 0; b < Kg; b++) row[b] = fmaf(av, ob[b], row[b]);
                }
            }
            } else {
                /* leaves of candidate a of group 2g: codebooks 2g*L .. 2g*L+L-1, entries tcur[...] */
                for (int a = 0; a < Kg; a++)
                    for (int b = 0; b < Kg; b++) {
                        const uint8_t *ta = tcur + ((size_t)(2 * g) * Kg + a) * L;
                        const uint8_t *tb = tcur + ((size_t)(2 * g + 1) * Kg + b) * L;
                        double tot = 0.0; float totf = 0.0f;
                        for (int j = 0; j < L; j++)
                            for (int j2 = 0; j2 < L; j2++) {
                                int ne = 2 * g * L + j, no = (2 * g + 1) * L + j2;
                                const float *ce = o->C + ((size_t)ne * K + ta[j]) * Dp;
                                const float *co = o->C + ((size_t)no * K + tb[j2]) * Dp;
                                const float *oe = s->old + (size_t)ne * Dp, *oo = s->old + (size_t)no * Dp;
                                double g1 = 0, g2 = 0, g3 = 0, g4 = 0;
                                if (g_pair_mode == 3) {
                                    float f1 = 0, f2 = 0, f3 = 0, f4 = 0;
                                    for (int i = 0; i < Dp; i++) {
                                        int d = o->order16[i];
                                        f1 = fmaf(ce[d], co[d], f1); f2 = fmaf(ce[d], oo[d], f2);
                                        f3 = fmaf(oe[d], co[d], f3); f4 = fmaf(oe[d], oo[d], f4);
                                    }
                                    totf = totf + (((f1 - f2) - f3) + f4);
                                    continue;
                                }
                                for (int d = 0; d < Dp; d++) {
                                    g1 += (double)ce[d] * co[d]; g2 += (double)ce[d] * oo[d];
                                    g3 += (double)oe[d] * co[d]; g4 += (double)oe[d] * oo[d];
                                }
                                if (g_pair_mode == 1) tot += ((g1 - g2) - g3) + g4;
                                else totf = totf + ((((float)g1 - (float)g2) - (float)g3) + (float)g4);
                            }
                        acc[(size_t)a * Kg + b] = (g_pair_mode == 1) ? (float)tot : totf;
                    }
            }
"""
assert old_pair in src
src = src.replace(old_pair, new_pair)

old_s0 = """        for (int i = 0; i < Dp; i++) {
            float xv = s->xrem[o->order16[i]];
            const float *row = o->CT + ((size_t)n * Dp + i) * K;
            for (int k = 0; k < K; k++) acc[k] = fmaf(row[k], xv, acc[k]);
        }
        const float *Q"""
new_s0 = """        if (g_s0_mode == 0) {
        for (int i = 0; i < Dp; i++) {
            float xv = s->xrem[o->order16[i]];
            const float *row = o->CT + ((size_t)n * Dp + i) * K;
            for (int k = 0; k < K; k++) acc[k] = fmaf(row[k], xv, acc[k]);
        }
        } else {
            /* XC = fp32 chain of c.x (one GEMM per encode); Gram sums over the other codebooks in fp64 */
            for (int i = 0; i < Dp; i++) {
                int d = o->order16[i];
                float xv = (d < D) ? x[d] : 0.0f;
                const float *row = o->CT + ((size_t)n * Dp + i) * K;
                for (int k = 0; k < K; k++) acc[k] = fmaf(row[k], xv, acc[k]);
            }
            for (int k = 0; k < K; k++) {
                const float *c = o->C + ((size_t)n * K + k) * Dp;
                double gs = 0.0; float gsf = 0.0f;
                for (int m = 0; m < N; m++) {
                    if (m == n) continue;
                    const float *om = s->old + (size_t)m * Dp;
                    double gd = 0.0;
                    if (g_s0_mode == 3) {
                        float f = 0.0f;
                        for (int i = 0; i < Dp; i++) { int d = o->order16[i]; f = fmaf(c[d], om[d], f); }
                        gsf = gsf + f;
                        continue;
                    }
                    for (int d = 0; d < Dp; d++) gd += (double)c[d] * om[d];
                    gs += (g_s0_mode == 2) ? (double)(float)gd : gd;
                }
                acc[k] = (g_s0_mode == 3) ? (gsf - acc[k]) : (float)(gs - (double)acc[k]);
            }
        }
        const float *Q"""
assert old_s0 in src
src = src.replace(old_s0, new_s0)
src = src.replace("static int round_up16(int d)", "static int g_pair_mode = 0, g_s0_mode = 0;\nvoid mcq_emu_modes(int p, int s0) { g_pair_mode = p; g_s0_mode = s0; }\nstatic int round_up16(int d)", 1)
open("/tmp/gram_emu2.c", "w").write(src)
subprocess.check_call("gcc -O3 -fPIC -shared -std=c11 -mfma -mavx2 -ffp-contract=off -fno-math-errno -fopenmp -I%s/oracle -o /tmp/libgram_emu2.so /tmp/gram_emu2.c %s/oracle/mcq_host.c -lm" % (ROOT, ROOT), shell=True)
print("built /tmp/libgram_emu2.so")
