"""CPU: numpy emulation of k_logmel's FFT index algebra (wakeword_trainer_home_amd/csrc/ww_frontend.hip):
1024 = 16 x 16 x 4 decomposition with 16 points per lane, the padded LDS slot map, the in-place pass 2,
the fft16 output-slot permutation, and pass 3 fused with the separation of two real frames packed into one complex FFT
(butterfly pairs that hold X[k] and X[N-k] in one lane's registers).
The HIP kernel transcribes exactly these index formulas; the GPU tests then check the arithmetic."""
import numpy as np

N, STRIDE = 1024, 68


def W(n, k, n_pts):
    return np.exp(-2j * np.pi * ((n * k) % n_pts) / n_pts)


def fft4(v0, v1, v2, v3):
    a, b, c, d = v0 + v2, v0 - v2, v1 + v3, v1 - v3
    return a + c, b - 1j * d, a - c, b + 1j * d


def fft16_slots(v):
    """v: list of 16 arrays in natural order -> list where X[k] sits at slot (k>>2) + 4*(k&3) (as in the kernel)."""
    v = list(v)
    for j1 in range(4):
        v[j1], v[j1 + 4], v[j1 + 8], v[j1 + 12] = fft4(v[j1], v[j1 + 4], v[j1 + 8], v[j1 + 12])
    for j1 in range(1, 4):
        for ka in range(1, 4):
            v[j1 + 4 * ka] = v[j1 + 4 * ka] * W(j1, ka, 16)
    for ka in range(4):
        v[4 * ka], v[4 * ka + 1], v[4 * ka + 2], v[4 * ka + 3] = fft4(v[4 * ka], v[4 * ka + 1], v[4 * ka + 2], v[4 * ka + 3])
    return v


def slot(k):
    return (k >> 2) + 4 * (k & 3)


def test_fft16_slot_permutation():
    rng = np.random.default_rng(0)
    x = rng.normal(size=16) + 1j * rng.normal(size=16)
    out = fft16_slots([np.array(t) for t in x])
    ref = np.fft.fft(x)
    for k in range(16):
        assert abs(out[slot(k)] - ref[k]) < 1e-12


def test_packed_1024_point_fft_index_algebra():
    rng = np.random.default_rng(1)
    xa, xb = rng.normal(size=N), rng.normal(size=N)
    lane = np.arange(64)
    buf = np.zeros(16 * STRIDE, complex)
    # pass 1: lane holds n = lane + 64 j ; radix-16 over j ; twiddle W1024^(lane*kb) ; store [kb][lane]
    regs = fft16_slots([xa[lane + 64 * j] + 1j * xb[lane + 64 * j] for j in range(16)])
    for kb in range(16):
        buf[kb * STRIDE + lane] = regs[slot(kb)] * W(lane, kb, 1024)
    # pass 2 (in place): lane = kb*4 + q reads [kb][4m+q], radix-16 over m, twiddle W64^(q*kc), writes [kb][4kc+q]
    kb2, q = lane >> 2, lane & 3
    regs = fft16_slots([buf[kb2 * STRIDE + 4 * m + q] for m in range(16)])
    touched_r = {(int(a), int(b)) for m in range(16) for a, b in zip(kb2, 4 * m + q)}
    touched_w = {(int(a), int(b)) for kc in range(16) for a, b in zip(kb2, 4 * kc + q)}
    assert touched_r == touched_w and len(touched_r) == 1024          # in place: each lane rewrites its own slots
    for kc in range(16):
        buf[kb2 * STRIDE + 4 * kc + q] = regs[slot(kc)] * W(16 * q * kc, 1, 1024)
    # the factors of passes 1 and 2 come from six table entries: W^(base*k) = W^(4*base*a) * W^(base*b), k = 4a + b
    for base in (int(lane[37]), 16 * 3):
        for kk in range(16):
            assert abs(W(base * 4 * (kk >> 2), 1, 1024) * W(base * (kk & 3), 1, 1024) - W(base * kk, 1, 1024)) < 1e-14
    # pass 3 fused with the separation: 128 units = pairs of butterflies (kb, kc) / (16-kb, 15-kc) whose outputs are
    # X[k] and X[N-k] (kd' = 3 - kd), plus one unit with the two self-paired butterflies (0,0) and (0,8)
    PB = np.full((2, 548), np.nan)
    written = []

    def emit(X_, Y_, kbin):
        j = kbin if kbin <= 512 else N - kbin
        pj = j + (j >> 4)
        assert pj < 548
        written.append(j)
        PB[0, pj] = 0.25 * ((X_.real + Y_.real) ** 2 + (X_.imag - Y_.imag) ** 2)
        PB[1, pj] = 0.25 * ((X_.imag + Y_.imag) ** 2 + (X_.real - Y_.real) ** 2)

    for idx in range(128):
        if idx < 112:
            kbA, kcA = 1 + (idx >> 4), idx & 15
            kbB, kcB = 16 - kbA, 15 - kcA
        elif idx < 120:
            kbA, kcA = 8, idx - 112
            kbB, kcB = 8, 15 - kcA
        elif idx < 127:
            kbA, kcA = 0, idx - 119
            kbB, kcB = 0, 16 - kcA
        else:
            kbA, kcA, kbB, kcB = 0, 0, 0, 8
        sa, sb = kbA * STRIDE + 4 * kcA, kbB * STRIDE + 4 * kcB
        oa = fft4(buf[sa], buf[sa + 1], buf[sa + 2], buf[sa + 3])
        ob = fft4(buf[sb], buf[sb + 1], buf[sb + 2], buf[sb + 3])
        if idx != 127:
            for kd in range(4):
                emit(oa[kd], ob[3 - kd], 16 * kcA + kbA + 256 * kd)
        else:
            emit(oa[0], oa[0], 0)
            emit(oa[1], oa[3], 256)
            emit(oa[2], oa[2], 512)
            emit(ob[0], ob[3], 128)
            emit(ob[1], ob[2], 384)
    assert sorted(written) == list(range(513))                         # every bin 0..512 exactly once
    j = np.arange(513)
    pa, pb = PB[0, j + (j >> 4)], PB[1, j + (j >> 4)]
    assert np.abs(pa - np.abs(np.fft.rfft(xa)) ** 2).max() < 1e-8
    assert np.abs(pb - np.abs(np.fft.rfft(xb)) ** 2).max() < 1e-8


def test_mel_band_halves_cover_each_band_once():
    from oracle.features import mel_filterbank
    fb = mel_filterbank(513, 40, 16000)
    for m in range(40):
        nz = np.nonzero(fb[:, m] > 0)[0]
        s, L = int(nz[0]), len(nz)
        assert np.array_equal(nz, np.arange(s, s + L))              # bands are contiguous
        h0 = (L + 1) >> 1
        halves = list(range(0, h0)) + list(range(h0, L))
        assert halves == list(range(L))


def test_gemm16_lds_image_swizzle_is_a_bijection_and_conflict_free():
    """ww_gemm16.hip: an operand tile (128 rows x 64 16-bit elements = 8 chunks of 16 B per row) is written by LDS-DMA, whose
    image is lane-linear (wave-instruction `piece` puts lane l at byte piece*1024 + 16*l), so the swizzle lives on the SOURCE
    side: lane l of piece p fetches chunk (l & 7) ^ ((row >> 1) & 7) of row = 8p + (l >> 3).  A fragment read of (row, chunk)
    goes to byte row*128 + 16*(chunk ^ ((row >> 1) & 7)).  Checked here: every (row, chunk) is found where the read looks for it,
    and each 16-lane service group of a ds_read_b128 (MI355X: {0-3,12-15,20-27}, {4-11,16-19,28-31}, +32) touches 16 different
    16-byte bank slots of the 256-byte LDS line for every fragment the kernel reads."""
    image = {}
    for piece in range(16):
        for lane_ in range(64):
            row = piece * 8 + (lane_ >> 3)
            chunk = (lane_ & 7) ^ ((row >> 1) & 7)
            image[piece * 1024 + 16 * lane_] = (row, chunk)
    assert len(image) == 128 * 8 and len(set(image.values())) == 128 * 8

    def addr(row, chunk):
        return row * 128 + 16 * (chunk ^ ((row >> 1) & 7))

    for row in range(128):
        for chunk in range(8):
            assert image[addr(row, chunk)] == (row, chunk)
    groups = [[0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27], [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31]]
    groups += [[l + 32 for l in g] for g in groups]
    for rowbase in (0, 32, 64, 96):
        for ks in range(4):
            for g in groups:
                slots = {(addr(rowbase + (l & 31), 2 * ks + (l >> 5)) % 256) // 16 for l in g}
                assert len(slots) == 16, (rowbase, ks, g)
