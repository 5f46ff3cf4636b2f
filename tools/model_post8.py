"""numpy model of post_fft8_kernel's resampler (kernels_post.hip): scipy.signal.resample(a, 2N) of a real sequence through ONE
half-size complex transform each way (real-input packing), cut as N/2 = 16 x LPI so that LPI lanes hold one item:
16-point transforms in the lanes' registers, one exchange, LPI-point transforms.  Index conventions and table layouts are
the kernel's; tests/test_host_logic.py checks the C table builder (sdrt_build_fft8_tables) against `tables()` and this
model against scipy."""
import numpy as np

P = 16          # points per lane


def lane_sets(LPI):
    """k1 values lane l finishes in the second forward step: closed under k1 -> (16 - k1) % 16, so that the partner
    Z[H - k] of every Z[k] the lane holds sits in the same lane (LPI = 8: two values per lane; 16: one, partner in lane 16 - l)."""
    if LPI == 8:
        return [[0, 8]] + [[l, 16 - l] for l in range(1, 8)]
    if LPI == 4:      # four values per lane: the pairs (l, 16 - l) and (4 + l, 12 - l); lane 0: 0 and 8 pair with themselves
        return [[l, (16 - l) if l else 8, 4 + l, 12 - l] for l in range(4)]
    return [[l] for l in range(16)]


def tables(N, LPI):
    """tw1[k1][l] = W_H^(l k1); tw2[ci][n2][l] = conj(W_H^(n2 c)), c = S_l[ci]; alpha/beta[ci][k2][l] for k = c + 16 k2."""
    H = N // 2
    assert H == P * LPI
    S = lane_sets(LPI)
    nc = len(S[0])
    W = lambda num, den: np.exp(-2j * np.pi * (num % den) / den)
    tw1 = np.array([[W(l * k1, H) for l in range(LPI)] for k1 in range(P)])
    tw2 = np.array([[[np.conj(W(n2 * S[l][ci], H)) for l in range(LPI)] for n2 in range(LPI)] for ci in range(nc)])
    sh = np.zeros(H + 1, complex)
    k = np.arange(H)
    sh[:H] = np.exp(1j * np.pi * k / N) / N          # half-sample shift / N; the Nyquist bin is dropped (scipy halves it twice)
    th = 2 * np.pi * k / N
    al = sh[k] * (1 - np.sin(th)) + np.conj(sh[H - k]) * (1 + np.sin(th))
    be = 1j * np.cos(th) * (sh[k] - np.conj(sh[H - k]))
    alpha = np.array([[[al[S[l][ci] + P * k2] for l in range(LPI)] for k2 in range(LPI)] for ci in range(nc)])
    beta = np.array([[[be[S[l][ci] + P * k2] for l in range(LPI)] for k2 in range(LPI)] for ci in range(nc)])
    return tw1, tw2, alpha, beta


def resample_odd(a, LPI):
    """odd output samples of scipy.signal.resample(a, 2 len(a)) by the kernel's steps"""
    N = len(a)
    H = N // 2
    S = lane_sets(LPI)
    nc = len(S[0])
    tw1, tw2, alpha, beta = tables(N, LPI)
    z = a[0::2] + 1j * a[1::2]
    x = z.reshape(P, LPI).T                            # x[l][k] = z[l + LPI k]
    X1 = np.fft.fft(x, axis=1) * tw1.T                 # [l][k1]
    # exchange: lane l takes V[c][n2] = X1[n2][c] for c in S_l
    Z = np.zeros(H, complex)
    for l in range(LPI):
        for ci in range(nc):
            c = S[l][ci]
            Z[c + P * np.arange(LPI)] = np.fft.fft(X1[:, c])          # over n2 -> k2
    Y = np.zeros((LPI, nc, LPI), complex)
    for l in range(LPI):
        for ci in range(nc):
            c = S[l][ci]
            for k2 in range(LPI):
                kk = c + P * k2
                Y[l, ci, k2] = alpha[ci, k2, l] * Z[kk] + beta[ci, k2, l] * np.conj(Z[(H - kk) % H])
    U = np.zeros((P, LPI), complex)                    # [c][n2]
    for l in range(LPI):
        for ci in range(nc):
            c = S[l][ci]
            U[c] = np.fft.ifft(Y[l, ci]) * LPI * tw2[ci, :, l]
    zz = np.fft.ifft(U, axis=0) * P                    # over c -> n1: zz[n1][n2] = z'[n2 + LPI n1]
    zp = zz.reshape(-1)
    b = np.empty(N)
    b[0::2] = zp.real
    b[1::2] = zp.imag
    return b


if __name__ == '__main__':
    from scipy.signal import resample
    rng = np.random.default_rng(1)
    for N, LPI in ((128, 4), (256, 8), (512, 16)):
        a = rng.uniform(-np.pi, np.pi, N)
        ref = resample(a, 2 * N)
        b = resample_odd(a, LPI)
        print(N, LPI, np.abs(ref[1::2] - b).max(), np.abs(ref[0::2] - a).max())
