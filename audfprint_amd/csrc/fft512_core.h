// fft512_core.h -- per-lane pieces of the 512-point complex FFT used by k_stft.
//
// One wavefront (64 lanes) transforms one complex sequence z[n] = xA[n] + i*xB[n] holding
// TWO real windowed frames; 512 = 8*8*8, so three in-register radix-8 butterflies with two
// exchanges through LDS in between (Stockham-style: every pass reads stride-1 per lane and
// the digit that was just transformed moves to the register index).
//
// Index algebra (W_N = exp(-2*pi*i/N)); n = 64*n2 + 8*n1 + n0, k = a + 8*b + 64*c:
//   pass 1  lane (n1,n0) = 8*n1+n0, register j=n2:  A[a]  = sum_n2 z[64 n2 + 8 n1 + n0] W_8^(n2 a)
//           twiddle A[a] *= W_64^(n1 a) W_512^(n0 a) = W_512^(L a)   (the second factor is pass 2's, hoisted)
//   xchg 1  lane (n1,n0) reg a  ->  lane (a,n0) = 8*a+n0, reg n1
//   pass 2  B[b] = sum_n1 A'[n1] W_8^(n1 b);  twiddle B[b] *= W_64^(n0 b)
//   xchg 2  lane (a,n0) reg b   ->  lane a+8*b, reg n0
//   pass 3  Z[a + 8 b + 64 c] = sum_n0 B'[n0] W_8^(n0 c)      -> lane m holds Z[m + 64 c], c = 0..7
//
// The functions are __host__ __device__ so tests/emul (host, no GPU) can run the very same
// arithmetic lane by lane (tests/test_fft_emulation.py builds tests/emul/fft_emul.cpp).
#pragma once

#if defined(__HIPCC__)
#define AFP_HD __host__ __device__ __forceinline__
#else
#define AFP_HD inline
#endif

// LDS layouts.  The exchanges move 8-byte elements (ds_write_b64 / ds_read_b64): the eight real parts of a lane, then
// the eight imaginary parts, through the same buffer of FFT_LDS_DOUBLES doubles per wavefront.
// xchg 1: writer lane L reg a -> a*72 + L;            reader lane L reg j -> (L>>3)*72 + 8*j + (L&7)
// xchg 2: writer lane L reg b -> (L&7)*66 + 8*b + (L>>3);  reader lane L reg j -> j*66 + L
// Row strides 72 / 66 (not 64) keep the 32-lane groups of ds_read_b64 on 32 distinct 8-byte slots (64 banks) and the
// 16-lane groups of ds_write_b64 on 16 distinct slots (32 banks): tests/test_fft_emulation.py audits both.
#define FFT_X1_STRIDE 72
#define FFT_X2_STRIDE 66
#define FFT_LDS_DOUBLES (8 * FFT_X1_STRIDE + 16)     // + 16: the Nyquist bins (re, im) of the wavefront's 8 frame pairs

AFP_HD int fft_x1_waddr(int lane, int a) { return a * FFT_X1_STRIDE + lane; }
AFP_HD int fft_x1_raddr(int lane, int j) { return (lane >> 3) * FFT_X1_STRIDE + 8 * j + (lane & 7); }
AFP_HD int fft_x2_waddr(int lane, int b) { return (lane & 7) * FFT_X2_STRIDE + 8 * b + (lane >> 3); }
AFP_HD int fft_x2_raddr(int lane, int j) { return j * FFT_X2_STRIDE + lane; }

// twiddle exponents (mod 512) this lane needs
// The textbook twiddles are W_64^(n1 a) after pass 1 and W_512^(n0 (a + 8 b)) after pass 2.  The b-independent factor
// W_512^(n0 a) of the second commutes with pass 2's DFT over n1 (it depends on (n0, a) only), so it is applied with the
// first: pass 1 multiplies by W_64^(n1 a) W_512^(n0 a) = W_512^(L a) (L = 8 n1 + n0 = the lane), pass 2 by
// W_64^(n0 b) -- register 0 of pass 2 needs no multiplication at all.
AFP_HD int fft_tw1_exp(int lane, int a) { return (lane * a) & 511; }                 // W_512^(L a)
AFP_HD int fft_tw2_exp(int lane, int b) { return (8 * (lane & 7) * b) & 511; }      // W_64^(n0 b) = W_512^(8 n0 b)

// In-place 8-point DFT, natural order in and out:  X[a] = sum_j x[j] W_8^(j a).
AFP_HD void dft8(double (&r)[8], double (&i)[8])
{
    const double h = 0.70710678118654752440;   // 1/sqrt(2)
    // radix-2 over (j, j+4)
    double sr0 = r[0] + r[4], si0 = i[0] + i[4], dr0 = r[0] - r[4], di0 = i[0] - i[4];
    double sr1 = r[1] + r[5], si1 = i[1] + i[5], dr1 = r[1] - r[5], di1 = i[1] - i[5];
    double sr2 = r[2] + r[6], si2 = i[2] + i[6], dr2 = r[2] - r[6], di2 = i[2] - i[6];
    double sr3 = r[3] + r[7], si3 = i[3] + i[7], dr3 = r[3] - r[7], di3 = i[3] - i[7];
    // odd branch inputs: d_j * W_8^j
    double er1 = (dr1 + di1) * h, ei1 = (di1 - dr1) * h;        // * (1 - i)/sqrt2
    double er2 = di2, ei2 = -dr2;                               // * (-i)
    double er3 = (di3 - dr3) * h, ei3 = -(dr3 + di3) * h;       // * (-1 - i)/sqrt2
    // DFT4 of s -> even outputs X[0], X[2], X[4], X[6]
    {
        double t0r = sr0 + sr2, t0i = si0 + si2, t1r = sr0 - sr2, t1i = si0 - si2;
        double t2r = sr1 + sr3, t2i = si1 + si3;
        double t3r = si1 - si3, t3i = -(sr1 - sr3);             // (s1 - s3) * (-i)
        r[0] = t0r + t2r; i[0] = t0i + t2i;
        r[4] = t0r - t2r; i[4] = t0i - t2i;
        r[2] = t1r + t3r; i[2] = t1i + t3i;
        r[6] = t1r - t3r; i[6] = t1i - t3i;
    }
    // DFT4 of e -> odd outputs X[1], X[3], X[5], X[7]
    {
        double t0r = dr0 + er2, t0i = di0 + ei2, t1r = dr0 - er2, t1i = di0 - ei2;
        double t2r = er1 + er3, t2i = ei1 + ei3;
        double t3r = ei1 - ei3, t3i = -(er1 - er3);
        r[1] = t0r + t2r; i[1] = t0i + t2i;
        r[5] = t0r - t2r; i[5] = t0i - t2i;
        r[3] = t1r + t3r; i[3] = t1i + t3i;
        r[7] = t1r - t3r; i[7] = t1i - t3i;
    }
}

AFP_HD void cmul(double& r, double& i, double wr, double wi)
{
    double nr = r * wr - i * wi;
    double ni = r * wi + i * wr;
    r = nr; i = ni;
}

// Power spectra of the two packed real frames from Z[k] = (zr, zi) and Z[512-k] = (pr, pi):
//   XA[k] = (Z[k] + conj Z[512-k]) / 2,   XB[k] = (Z[k] - conj Z[512-k]) / (2i)
AFP_HD void split_power(double zr, double zi, double pr, double pi, double& pa, double& pb)
{
    double ar = zr + pr, ai = zi - pi;
    double br = zr - pr, bi = zi + pi;
    pa = 0.25 * (ar * ar + ai * ai);
    pb = 0.25 * (br * br + bi * bi);
}
