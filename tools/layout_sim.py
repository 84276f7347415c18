"""CPU model of the index arithmetic of the 32x32x16 GEMM paths (gemm.hip, round 5): LDS-DMA placement -> swizzled
fragment reads -> MFMA lane maps -> epilogue32 addresses, carried out on LABELS instead of numbers. There is no GPU in
the authoring container: this is how the address math of a new main loop is checked before it costs a GPU call.
For every output element a lane stores, the model checks that the products the MFMAs accumulated there are exactly
X[m][k] * W[c][k] for all k of the K-tile, and that each ds_read_b128 lane group touches 16 distinct 16-byte slots of
the 256-byte bank row (conflict-free). Run: python tools/layout_sim.py"""
import itertools


def key(row):
    return ((row >> 1) ^ (row >> 4)) & 7


def perm32(v):
    return (((v >> 2) & 1) << 4) | ((v >> 3) << 2) | (v & 3)


B128_GROUPS = [[0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27],
               [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31],
               [32, 33, 34, 35, 44, 45, 46, 47, 52, 53, 54, 55, 56, 57, 58, 59],
               [36, 37, 38, 39, 40, 41, 42, 43, 48, 49, 50, 51, 60, 61, 62, 63]]


def check_banks(addrs, what):
    """addrs[lane] = LDS byte address of a ds_read_b128: every lane group must hit 16 distinct 16-byte slots mod 256 B"""
    for grp in B128_GROUPS:
        slots = {(addrs[l] >> 4) & 15 for l in grp}
        assert len(slots) == 16, (what, sorted(slots))


def dma_image(threads, rows, src_row_of):
    """LDS image of a [rows][64 k] operand tile written by glds_offsets*: chunk P = i*threads + tid holds source
    (row src_row_of(P >> 3), 16-byte k-chunk (P & 7) ^ key(P >> 3)). Returns img[byte address >> 4] = (src row, k chunk)."""
    img = {}
    for P in range(rows * 8):
        r, cs = P >> 3, P & 7
        img[P] = (src_row_of(r), cs ^ key(r))
    return img


def frag32(img, base_row, lane, ks, rowblk_key):
    """what gemm.hip reads for a 32x32x16 operand fragment: row base_row + l31, chunk (2 ks + hh) ^ k32 ^ rowblk_key"""
    l31, hh = lane & 31, lane >> 5
    k32 = ((l31 >> 1) ^ (l31 >> 4)) & 7
    addr = (base_row + l31) * 128 + ((((2 * ks + hh) ^ k32 ^ rowblk_key) & 7) << 4)
    return addr, img[addr >> 4]


def mfma32_labels(a, b):
    """a[lane] = (row label, k chunk) of the a-operand, b[lane] likewise. Returns D[lane][r] = (i label, j label, set of k chunks)
    for D[i][j] += sum_k A[i][k] B[k][j]: lane l supplies A[i = l&31][k chunk l>>5], B[k chunk l>>5][j = l&31]; lane l register r
    holds D[i = (r&3) + 8(r>>2) + 4(l>>5)][j = l&31]."""
    A = {}
    Bm = {}
    for l in range(64):
        A[(l & 31, l >> 5)] = a[l]
        Bm[(l >> 5, l & 31)] = b[l]
    out = {}
    for l in range(64):
        for r in range(16):
            i, j = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5), l & 31
            ks = set()
            rows_i, rows_j = set(), set()
            for kc in range(2):
                (ri, ka), (rj, kb) = A[(i, kc)], Bm[(kc, j)]
                assert ka == kb, "a and b fragments disagree on the contraction chunk"
                ks.add(ka)
                rows_i.add(ri)
                rows_j.add(rj)
            assert len(rows_i) == 1 and len(rows_j) == 1
            out[(l, r)] = (rows_i.pop(), rows_j.pop(), ks)
    return out


def sim_128x128():
    """gemm_kernel<..., PERM, MF32>: 4 waves (wm, wn) of 64 x 64; A tile rows natural, B tile rows perm32 per 32-row block"""
    imgA = dma_image(256, 128, lambda r: r)
    imgB = dma_image(256, 128, lambda r: (r & ~31) + perm32(r & 31))
    for wm, wn in itertools.product(range(2), range(2)):
        got = {}
        for ks in range(4):
            for rb, nb in itertools.product(range(2), range(2)):
                a_addr, b_addr, a, b = [], [], [], []
                for lane in range(64):
                    ad, lab = frag32(imgB, wn * 64 + nb * 32, lane, ks, wn * 4 + nb * 2)
                    a_addr.append(ad); a.append(lab)
                    ad, lab = frag32(imgA, wm * 64 + rb * 32, lane, ks, wm * 4 + rb * 2)
                    b_addr.append(ad); b.append(lab)
                check_banks(a_addr, "128 a"); check_banks(b_addr, "128 b")
                for (l, r), (ci, rj, kset) in mfma32_labels(a, b).items():
                    e = got.setdefault((rb, nb, l, r), [ci, rj, set()])
                    assert e[0] == ci and e[1] == rj
                    assert not (e[2] & kset)
                    e[2] |= kset
        # epilogue32: lane (l31, h) register r of acc[rb][nb] -> C[wm*64 + rb*32 + l31][wn*64 + nb*32 + 16h + r]
        for (rb, nb, l, r), (ci, rj, kset) in got.items():
            assert kset == set(range(8))
            assert rj == wm * 64 + rb * 32 + (l & 31), (rj, wm, rb, l)
            assert ci == wn * 64 + nb * 32 + 16 * (l >> 5) + r, (ci, wn, nb, l, r)
    print("128x128 MF32: fragments, bank slots and epilogue32 columns consistent")


def sim_256_8wave():
    """gemm_nt_256_kernel<., MF32>: half-tile images Amq_h (h = 0, 1) and Bnq_h of 128 rows; 8 waves (wr, wc)"""
    for h_a, h_b in itertools.product(range(2), range(2)):  # the quadrant (mq = h_a, nq = h_b) of every wave
        imgA = dma_image(512, 128, lambda r: (r >> 6) * 128 + h_a * 64 + (r & 63))
        imgB = dma_image(512, 128, lambda r: (r >> 5) * 64 + h_b * 32 + perm32(r & 31))
        for wr, wc in itertools.product(range(2), range(4)):
            got = {}
            for ks in range(4):
                for b_ in range(2):
                    a_addr, b_addr, a, b = [], [], [], []
                    for lane in range(64):
                        ad, lab = frag32(imgB, wc * 32, lane, ks, wc * 2)
                        a_addr.append(ad); a.append(lab)
                        ad, lab = frag32(imgA, wr * 64 + b_ * 32, lane, ks, wr * 4 + b_ * 2)
                        b_addr.append(ad); b.append(lab)
                    check_banks(a_addr, "256 a"); check_banks(b_addr, "256 b")
                    for (l, r), (ci, rj, kset) in mfma32_labels(a, b).items():
                        e = got.setdefault((b_, l, r), [ci, rj, set()])
                        assert e[0] == ci and e[1] == rj and not (e[2] & kset)
                        e[2] |= kset
            # store_quadrant(mq): epilogue32(acc32[mq], row0 + wr*128 + mq*64, col0 + wc*64): acc32[mq][b][nq]
            for (b_, l, r), (ci, rj, kset) in got.items():
                assert kset == set(range(8))
                assert rj == wr * 128 + h_a * 64 + b_ * 32 + (l & 31)
                assert ci == wc * 64 + h_b * 32 + 16 * (l >> 5) + r
    print("256x256 8-wave MF32: fragments, bank slots and epilogue32 columns consistent")


def sim_256_4wave():
    """gemm_nt_w4_kernel: whole-tile images A [256 rows] and B [256 rows, perm32 per 32-row block]; 4 waves (wr, wc) of 128 x 128
    = 4 x 4 blocks; fragment (block b of the wave, K-step ks): row w*128 + b*32 + l31, chunk (2ks + hh) ^ k32 ^ ((b*2) & 7)"""
    imgA = dma_image(256, 256, lambda r: r)
    imgB = dma_image(256, 256, lambda r: (r & ~31) + perm32(r & 31))
    for wr, wc in itertools.product(range(2), range(2)):
        got = {}
        for ks in range(4):
            for rb, nb in itertools.product(range(4), range(4)):
                a_addr, b_addr, a, b = [], [], [], []
                for lane in range(64):
                    ad, lab = frag32(imgB, wc * 128 + nb * 32, lane, ks, (wc * 8 + nb * 2) & 7)
                    a_addr.append(ad); a.append(lab)
                    ad, lab = frag32(imgA, wr * 128 + rb * 32, lane, ks, (wr * 8 + rb * 2) & 7)
                    b_addr.append(ad); b.append(lab)
                check_banks(a_addr, "w4 a"); check_banks(b_addr, "w4 b")
                for (l, r), (ci, rj, kset) in mfma32_labels(a, b).items():
                    e = got.setdefault((rb, nb, l, r), [ci, rj, set()])
                    assert e[0] == ci and e[1] == rj and not (e[2] & kset)
                    e[2] |= kset
        for (rb, nb, l, r), (ci, rj, kset) in got.items():
            assert kset == set(range(8))
            assert rj == wr * 128 + rb * 32 + (l & 31)
            assert ci == wc * 128 + nb * 32 + 16 * (l >> 5) + r
    print("256x256 4-wave MF32: fragments, bank slots and epilogue32 columns consistent")


if __name__ == "__main__":
    sim_128x128()
    sim_256_8wave()
    sim_256_4wave()
