// Host emulation of k_stft's per-wavefront FFT data flow (64 lanes x 8 registers, two LDS
// exchanges, partner shuffle, split of the two packed real frames).  Uses the SAME
// fft512_core.h the kernel uses, so the index algebra, twiddle exponents, LDS addressing and
// butterfly arithmetic are validated on the CPU against numpy.fft.rfft (tests/test_fft_emulation.py).
#include <cmath>
#include <cstring>
#include <vector>
#include "../../audfprint_amd/csrc/fft512_core.h"

// gen != 0: twiddles the way the kernel forms them (k_stft.hip, STFT_LOWREG): only the generators come from the
// table -- W_64^n1 for pass 1, W_512^(n0 a) and W_512^(8 n0) for pass 2 -- and the others are repeated complex
// products; gen == 0: every twiddle straight from the table.
static int emul_pair(const double* xa, const double* xb, double* pa, double* pb, int gen)
{
    // twiddle table exactly as the library builds it (afp_abi.hip: make_twiddles)
    std::vector<double> tw(1024);
    for (int m = 0; m < 512; m++) {
        long double ang = -2.0L * 3.14159265358979323846264338327950288L * m / 512.0L;
        tw[2 * m] = (double)cosl(ang);
        tw[2 * m + 1] = (double)sinl(ang);
    }
    static double R[64][8], I[64][8];
    std::vector<double> lr(FFT_LDS_DOUBLES), li(FFT_LDS_DOUBLES);
    for (int l = 0; l < 64; l++)
        for (int j = 0; j < 8; j++) { R[l][j] = xa[l + 64 * j]; I[l][j] = xb[l + 64 * j]; }
    // pass 1
    for (int l = 0; l < 64; l++) {
        dft8(R[l], I[l]);
        if (gen) {
            const int e1 = fft_tw1_exp(l, 1);
            double wr = tw[2 * e1], wi = tw[2 * e1 + 1];
            for (int a = 1; a < 8; a++) { cmul(R[l][a], I[l][a], wr, wi); if (a < 7) cmul(wr, wi, tw[2 * e1], tw[2 * e1 + 1]); }
        } else
        for (int a = 1; a < 8; a++) { int e = fft_tw1_exp(l, a); cmul(R[l][a], I[l][a], tw[2 * e], tw[2 * e + 1]); }
        for (int a = 0; a < 8; a++) { lr[fft_x1_waddr(l, a)] = R[l][a]; li[fft_x1_waddr(l, a)] = I[l][a]; }
    }
    for (int l = 0; l < 64; l++)
        for (int j = 0; j < 8; j++) { R[l][j] = lr[fft_x1_raddr(l, j)]; I[l][j] = li[fft_x1_raddr(l, j)]; }
    // pass 2
    for (int l = 0; l < 64; l++) {
        dft8(R[l], I[l]);
        if (gen) {
            const int es = fft_tw2_exp(l, 1);
            double wr = tw[2 * es], wi = tw[2 * es + 1];
            for (int b = 1; b < 8; b++) { cmul(R[l][b], I[l][b], wr, wi); if (b < 7) cmul(wr, wi, tw[2 * es], tw[2 * es + 1]); }
        } else
        for (int b = 1; b < 8; b++) { int e = fft_tw2_exp(l, b); cmul(R[l][b], I[l][b], tw[2 * e], tw[2 * e + 1]); }
    }
    for (int l = 0; l < 64; l++)
        for (int b = 0; b < 8; b++) { lr[fft_x2_waddr(l, b)] = R[l][b]; li[fft_x2_waddr(l, b)] = I[l][b]; }
    for (int l = 0; l < 64; l++)
        for (int j = 0; j < 8; j++) { R[l][j] = lr[fft_x2_raddr(l, j)]; I[l][j] = li[fft_x2_raddr(l, j)]; }
    // pass 3
    for (int l = 0; l < 64; l++) dft8(R[l], I[l]);
    // partner fetch + split (k_stft: P[c] = reg[7-c] of lane (64-m)&63; lane 0 remaps)
    for (int m = 0; m < 64; m++) {
        int pl = (64 - m) & 63;
        double Pr[4], Pi[4];
        for (int c = 0; c < 4; c++) { Pr[c] = R[pl][7 - c]; Pi[c] = I[pl][7 - c]; }
        for (int c = 0; c < 4; c++) {
            double qr, qi;
            if (m == 0) { qr = (c == 0) ? R[0][0] : Pr[c - 1]; qi = (c == 0) ? I[0][0] : Pi[c - 1]; }
            else { qr = Pr[c]; qi = Pi[c]; }
            split_power(R[m][c], I[m][c], qr, qi, pa[m + 64 * c], pb[m + 64 * c]);
        }
        if (m == 0) split_power(R[0][4], I[0][4], R[0][4], I[0][4], pa[256], pb[256]);
    }
    return 0;
}

extern "C" int emul_stft_pair(const double* xa, const double* xb, double* pa, double* pb) { return emul_pair(xa, xb, pa, pb, 0); }
extern "C" int emul_stft_pair_gen(const double* xa, const double* xb, double* pa, double* pb) { return emul_pair(xa, xb, pa, pb, 1); }
