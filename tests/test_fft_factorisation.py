"""CPU: the index arithmetic of frontend_fft_kernel (msk144cudecoder_amd/csrc/frontend.hip; Analytic::execute, analytic_fft.cu:84-157) in numpy.

The kernel does the reference's forward FFT - spectral mask - inverse FFT pair as a 16 x 16 x 32 mixed-radix decomposition,
n = 512 n1 + 32 n2 + n3, k = k1 + 16 k2 + 256 k3: after the three forward passes bin k sits at cell 512 k1 + 32 k2 + k3, the mask is applied
there (negative frequencies = k3 >= 16), and the inverse walks the factorisation backwards to natural order.  This test runs the same
passes, cell for cell and twiddle for twiddle, with numpy's FFT for the small transforms, and checks them against numpy's 8192-point FFT;
the in-register radix-2 butterflies (fft_reg) and their literal W32 twiddles are restated and checked the same way.  The GPU tests then hold
the kernel itself to 1e-5 of the window rms against the oracle (tests/test_gpu_parity.py)."""
import numpy as np

N = 8192
W = np.exp(-2j * np.pi * np.arange(N) / N)


def small_dft(a, inverse=False):
    return np.fft.ifft(a) * len(a) if inverse else np.fft.fft(a)


def kernel_passes(x, weight):
    """x: 8192 complex (zero-padded window); weight[k]: real spectral weight of bin k (0 for k >= 4096)."""
    s = x.astype(np.complex128).copy()
    out = np.zeros(N, complex)
    for t in range(512):                                   # forward, radix 16 over n1, twiddle W^(t k1)
        a = small_dft(s[np.arange(16) * 512 + t])
        out[np.arange(16) * 512 + t] = a * W[(t * np.arange(16)) % N]
    s = out
    for b in range(512):                                   # forward, radix 16 over n2, twiddle W512^(n3 k2) = W^(16 n3 k2)
        k1, n3 = b >> 5, b & 31
        idx = k1 * 512 + np.arange(16) * 32 + n3
        s[idx] = small_dft(s[idx]) * W[(16 * n3 * np.arange(16)) % N]
    bins = np.zeros(N, complex)
    for b in range(256):                                   # forward radix 32 over n3 | mask | inverse radix 32 over k3, one column per thread
        k1, k2 = b >> 4, b & 15
        a = small_dft(s[b * 32:(b + 1) * 32])
        k = k1 + 16 * k2 + 256 * np.arange(32)
        bins[k] = a
        a = a * weight[k]
        u = small_dft(a, inverse=True) * np.conj(W[(16 * np.arange(32) * k2) % N])
        s[b * 32:(b + 1) * 32] = u
    for b in range(512):                                   # inverse, radix 16 over k2, twiddle conj W^((32 n2 + n3) k1)
        k1, n3 = b >> 5, b & 31
        idx = k1 * 512 + np.arange(16) * 32 + n3
        s[idx] = small_dft(s[idx], inverse=True) * np.conj(W[((np.arange(16) * 32 + n3) * k1) % N])
    y = np.zeros(N, complex)
    for t in range(512):                                   # inverse, radix 16 over k1: natural order
        y[np.arange(16) * 512 + t] = small_dft(s[np.arange(16) * 512 + t], inverse=True)
    return bins, y


def test_three_passes_put_bin_k_where_the_mask_expects_it_and_the_inverse_ends_in_natural_order():
    rng = np.random.default_rng(5)
    x = np.zeros(N, complex)
    x[:5184] = rng.normal(size=5184)                       # real, zero-padded: what the front end feeds
    k = np.arange(N)
    weight = np.where(k < N // 2, 1.0 / (1 + k % 7), 0.0)  # any weight on the lower half, the upper half zeroed
    weight[0] *= 0.5                                        # half DC (analytic_fft.cu:124)
    bins, y = kernel_passes(x, weight)
    X = np.fft.fft(x)
    assert np.abs(bins - X).max() < 1e-9 * np.abs(X).max()
    ref = np.fft.ifft(X * weight) * N                       # cuFFT's inverse is unnormalised
    assert np.abs(y - ref).max() < 1e-9 * np.abs(ref).max()


def _cos_pi16(k):
    return np.cos(k * np.pi / 16)


def _mul_w32(x, k, inverse):
    if k == 0:
        return x
    if k == 8:
        return complex(-x.imag, x.real) if inverse else complex(x.imag, -x.real)
    c = _cos_pi16(k)
    sn = _cos_pi16(8 - k if k < 8 else k - 8)               # sin(k pi / 16)
    s = sn if inverse else -sn
    return complex(x.real * c - x.imag * s, x.real * s + x.imag * c)


def fft_reg(a, inverse):
    """frontend.hip fft_reg<R, kInverse>: radix-2 decimation-in-frequency stages on a register array, then the bit reversal."""
    a = list(a)
    R = len(a)
    log = {16: 4, 32: 5}[R]
    for stage in range(log):
        ln = R >> stage
        half = ln >> 1
        for b in range(0, R, ln):
            for j in range(half):
                u, v = a[b + j], a[b + j + half]
                a[b + j] = u + v
                a[b + j + half] = _mul_w32(u - v, j * (32 // ln), inverse)
    for i in range(R):
        r = 0
        for bit in range(log):
            r |= ((i >> bit) & 1) << (log - 1 - bit)
        if i < r:
            a[i], a[r] = a[r], a[i]
    return np.array(a)


def test_register_butterflies_are_the_small_dfts():
    rng = np.random.default_rng(6)
    for R in (16, 32):
        x = rng.normal(size=R) + 1j * rng.normal(size=R)
        assert np.abs(fft_reg(x, False) - np.fft.fft(x)).max() < 1e-12
        assert np.abs(fft_reg(x, True) - np.fft.ifft(x) * R).max() < 1e-12


def test_padded_cells_are_conflict_free():
    """cell(i) = i + i // 32: the radix-32 columns are 33 cells apart, so the 32 lanes of a ds_read_b64 group (one float2 each) reading
    element n3 of 32 consecutive columns hit 32 different bank pairs; the stride-512 and stride-32 passes read consecutive cells."""
    cell = lambda i: i + (i >> 5)
    for n3 in range(32):
        banks = {(2 * cell(b * 32 + n3)) % 64 for b in range(32)}          # float2 = 2 dwords, 64 banks
        assert len(banks) == 32
    for n1 in range(16):
        lanes = [cell(n1 * 512 + t) for t in range(32)]
        assert lanes == list(range(lanes[0], lanes[0] + 32))
    assert cell(8191) < 8192 + 256                                         # kFftCells
