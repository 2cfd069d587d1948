"""Pure-Python big-int model of the BLS12-377 / BLS12-381 fields and curves.

Build-time tooling only (constant generation + small golden vectors); never on the
product path.  Constants follow SURVEY.md §A.1 and are re-derived / re-checked here.
"""
import random

X377 = 0x8508c00000000001
R377 = X377**4 - X377**2 + 1
Q377 = ((X377 - 1)**2 * R377) // 3 + X377
assert R377 == 0x12ab655e9a2ca55660b44d1e5c37b00159aa76fed00000010a11800000000001
assert Q377 == 0x01ae3a4617c510eac63b05c06ca1493b1a22d9f300f5138f1ef3622fba094800170b5d44300000008508c00000000001

X381 = -0xd201000000010000
R381 = X381**4 - X381**2 + 1
Q381 = ((X381 - 1)**2 * R381) // 3 + X381
assert R381 == 0x73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001
assert Q381 == 0x1a0111ea397fe69a4b1ba7b6434bacd764774b84f38512bf6730d2a0f6b0f6241eabfffeb153ffffb9feffffffffaaab

G1_377 = (0x008848defe740a67c8fc6225bf87ff5485951e2caa9d41bb188282c8bd37cb5cd5481512ffcd394eeab9b16eb21be9ef,
          0x01914a69c5102eff1f674f5d30afeec4bd7fb348ca3e52d96d182ad44fb82305c2fe3d3634a9591afd82de55559c8ea6)
G1_381 = (0x17f1d3a73197d7942695638c4fa9ac0fc3688c4f9774b905a14e3a3f171bac586c55e83ff97a1aeffb3af00adb22c6bb,
          0x08b3f481e3aaa0f1a09e30ed741d8ae4fcf5e095d5d00af600db18cb2c04b3edd03cc744a2888ae40caa232946c5e7e1)
assert (G1_377[1]**2 - G1_377[0]**3 - 1) % Q377 == 0
assert (G1_381[1]**2 - G1_381[0]**3 - 4) % Q381 == 0


def two_adicity(p):
    s, t = 0, p - 1
    while t % 2 == 0:
        s += 1
        t //= 2
    return s, t


def mont_consts(p, nbits):
    R = (1 << nbits) % p
    R2 = R * R % p
    inv64 = (-pow(p, -1, 1 << 64)) % (1 << 64)
    inv32 = (-pow(p, -1, 1 << 32)) % (1 << 32)
    return R, R2, inv64, inv32


FR = {
    "377": dict(p=R377, gen=22, bits=253),
    "381": dict(p=R381, gen=7, bits=255),
}
for k, f in FR.items():
    s, t = two_adicity(f["p"])
    f["two_adicity"] = s
    f["root"] = pow(f["gen"], t, f["p"])
    assert pow(f["root"], 1 << s, f["p"]) == 1 and pow(f["root"], 1 << (s - 1), f["p"]) != 1
assert FR["377"]["two_adicity"] == 47 and FR["381"]["two_adicity"] == 32
assert FR["377"]["root"] == 8065159656716812877374967518403273466521432693661810619979959746626482506078
assert FR["381"]["root"] == 10238227357739495823651030575849232062558860180284477541189508159991286009131


# ---------------- generic short-Weierstrass arithmetic over Fp (affine, None = infinity) -------------
def ec_add(P, Q, p):
    if P is None:
        return Q
    if Q is None:
        return P
    x1, y1 = P
    x2, y2 = Q
    if x1 == x2:
        if (y1 + y2) % p == 0:
            return None
        lam = 3 * x1 * x1 * pow(2 * y1, -1, p) % p
    else:
        lam = (y2 - y1) * pow(x2 - x1, -1, p) % p
    x3 = (lam * lam - x1 - x2) % p
    return (x3, (lam * (x1 - x3) - y1) % p)


def ec_mul(k, P, p):
    R = None
    while k:
        if k & 1:
            R = ec_add(R, P, p)
        P = ec_add(P, P, p)
        k >>= 1
    return R


# ---------------- Fq2 = Fq[u]/(u^2 - NR) -------------
class Fq2:
    def __init__(self, p, nr):
        self.p, self.nr = p, nr % p

    def add(self, a, b):
        return ((a[0] + b[0]) % self.p, (a[1] + b[1]) % self.p)

    def sub(self, a, b):
        return ((a[0] - b[0]) % self.p, (a[1] - b[1]) % self.p)

    def mul(self, a, b):
        p = self.p
        return ((a[0] * b[0] + self.nr * a[1] * b[1]) % p, (a[0] * b[1] + a[1] * b[0]) % p)

    def inv(self, a):
        p = self.p
        n = pow((a[0] * a[0] - self.nr * a[1] * a[1]) % p, -1, p)
        return (a[0] * n % p, (-a[1]) * n % p)

    def pow(self, a, e):
        r = (1, 0)
        while e:
            if e & 1:
                r = self.mul(r, a)
            a = self.mul(a, a)
            e >>= 1
        return r

    def sqrt(self, a):
        """Square root in Fq2 by generic Tonelli-Shanks over the group of order p^2-1 (or None)."""
        p = self.p
        if a == (0, 0):
            return (0, 0)
        order = p * p - 1
        if self.pow(a, order // 2) != (1, 0):
            return None
        s, t = 0, order
        while t % 2 == 0:
            s += 1
            t //= 2
        rnd = random.Random(7)
        while True:
            z = (rnd.randrange(p), rnd.randrange(p))
            if z != (0, 0) and self.pow(z, order // 2) != (1, 0):
                break
        c = self.pow(z, t)
        x = self.pow(a, (t + 1) // 2)
        b = self.pow(a, t)
        m = s
        while b != (1, 0):
            i, bb = 0, b
            while bb != (1, 0):
                bb = self.mul(bb, bb)
                i += 1
            g = c
            for _ in range(m - i - 1):
                g = self.mul(g, g)
            x = self.mul(x, g)
            c = self.mul(g, g)
            b = self.mul(b, c)
            m = i
        return x


def ec2_add(F, P, Q):
    if P is None:
        return Q
    if Q is None:
        return P
    x1, y1 = P
    x2, y2 = Q
    if x1 == x2:
        if F.add(y1, y2) == (0, 0):
            return None
        lam = F.mul(F.mul((3, 0), F.mul(x1, x1)), F.inv(F.add(y1, y1)))
    else:
        lam = F.mul(F.sub(y2, y1), F.inv(F.sub(x2, x1)))
    x3 = F.sub(F.sub(F.mul(lam, lam), x1), x2)
    return (x3, F.sub(F.mul(lam, F.sub(x1, x3)), y1))


def ec2_mul(F, k, P):
    R = None
    while k:
        if k & 1:
            R = ec2_add(F, R, P)
        P = ec2_add(F, P, P)
        k >>= 1
    return R


def derive_g2_377():
    """BLS12-377: Fq2 = Fq[u]/(u^2+5); Fq6 = Fq2[v]/(v^3-u); D-type twist E': y^2 = x^3 + 1/u.
    Returns (b_twist, generator of the order-r subgroup of E'(Fq2), cofactor)."""
    p, r = Q377, R377
    F = Fq2(p, -5)
    bt = F.inv((0, 1))
    t = X377 + 1                       # trace of Frobenius of E/Fq
    assert (p + 1 - t) % r == 0
    t2 = t * t - 2 * p                 # trace over Fq2
    # 4 p^2 - t2^2 = 3 f^2  (CM discriminant -3)
    from math import isqrt
    f2 = isqrt((4 * p * p - t2 * t2) // 3)
    assert 3 * f2 * f2 == 4 * p * p - t2 * t2
    cands = [p * p + 1 - (t2 + 3 * f2) // 2, p * p + 1 - (t2 - 3 * f2) // 2,
             p * p + 1 + (t2 + 3 * f2) // 2, p * p + 1 + (t2 - 3 * f2) // 2]
    # find a point on E'
    xi = 1
    while True:
        x = (xi, 1)
        y = F.sqrt(F.add(F.mul(F.mul(x, x), x), bt))
        if y is not None:
            break
        xi += 1
    P = (x, y)
    order = None
    for n in cands:
        if n % r == 0 and ec2_mul(F, n, P) is None:
            order = n
            break
    assert order is not None
    cof = order // r
    G = ec2_mul(F, cof, P)
    assert G is not None and ec2_mul(F, r, G) is None
    return bt, G, cof


if __name__ == "__main__":
    bt, G, cof = derive_g2_377()
    print("twist b =", bt)
    print("G2 gen =", G)


# ---------------- ark_std::test_rng() + arkworks 0.3 sampling order of KZG10::setup [RECALL] -------------
# rand 0.8 StdRng = ChaCha12 through rand_core's BlockRng with a 64-word buffer; ark-ff UniformRand for Fp: limbs from next_u64, shave the
# unused top bits, reject >= p, the limbs ARE the Montgomery representation; ark-ec GroupProjective::rand: x = BaseField::rand,
# greatest = rng.gen::<bool>() (top bit of one next_u32), get_point_from_x (pick y by (y < -y) ^ greatest), scale_by_cofactor.
ARK_TEST_RNG_SEED = bytes([1, 0, 0, 0, 23, 0, 0, 0, 200, 1, 0, 0, 210, 30, 0, 0] + [0] * 16)


class StdRngModel:
    def __init__(self, seed=ARK_TEST_RNG_SEED, rounds=12):
        self.key = [int.from_bytes(seed[4 * i:4 * i + 4], "little") for i in range(8)]
        self.rounds, self.counter, self.buf, self.idx = rounds, 0, [], 64

    def _block(self, ctr):
        M = 0xFFFFFFFF
        s = [0x61707865, 0x3320646e, 0x79622d32, 0x6b206574] + self.key + [ctr & M, ctr >> 32, 0, 0]
        x = list(s)

        def rol(v, n):
            return ((v << n) | (v >> (32 - n))) & M

        def Q(a, b, c, d):
            x[a] = (x[a] + x[b]) & M; x[d] = rol(x[d] ^ x[a], 16); x[c] = (x[c] + x[d]) & M; x[b] = rol(x[b] ^ x[c], 12)
            x[a] = (x[a] + x[b]) & M; x[d] = rol(x[d] ^ x[a], 8); x[c] = (x[c] + x[d]) & M; x[b] = rol(x[b] ^ x[c], 7)
        for _ in range(self.rounds // 2):
            Q(0, 4, 8, 12); Q(1, 5, 9, 13); Q(2, 6, 10, 14); Q(3, 7, 11, 15)
            Q(0, 5, 10, 15); Q(1, 6, 11, 12); Q(2, 7, 8, 13); Q(3, 4, 9, 14)
        return [(x[i] + s[i]) & M for i in range(16)]

    def _refill(self):
        self.buf = sum((self._block(self.counter + b) for b in range(4)), [])
        self.counter += 4
        self.idx = 0

    def next_u32(self):
        if self.idx >= 64:
            self._refill()
        v = self.buf[self.idx]
        self.idx += 1
        return v

    def next_u64(self):
        if self.idx < 63:
            lo, hi = self.buf[self.idx], self.buf[self.idx + 1]
            self.idx += 2
        elif self.idx >= 64:
            self._refill()
            lo, hi = self.buf[0], self.buf[1]
            self.idx = 2
        else:
            lo = self.buf[63]
            self._refill()
            hi = self.buf[0]
            self.idx = 1
        return (hi << 32) | lo

    def rand_fp(self, p, limbs64):
        """returns the VALUE of the sampled element (the sampled limbs are its Montgomery representation)"""
        bits = p.bit_length()
        R = 1 << (64 * limbs64)
        while True:
            v = 0
            for i in range(limbs64):
                v |= self.next_u64() << (64 * i)
            v &= (1 << bits) - 1                     # shave the top limb down to the modulus bit length
            if v < p:
                return v * pow(R, -1, p) % p

    def rand_bool(self):
        return (self.next_u32() >> 31) == 1


def fp_sqrt(a, p):
    """Tonelli-Shanks in Fp (any root, or None)"""
    a %= p
    if a == 0:
        return 0
    if pow(a, (p - 1) // 2, p) != 1:
        return None
    s, t = 0, p - 1
    while t % 2 == 0:
        s += 1
        t //= 2
    z = 2
    while pow(z, (p - 1) // 2, p) != p - 1:
        z += 1
    c, x, b, m = pow(z, t, p), pow(a, (t + 1) // 2, p), pow(a, t, p), s
    while b != 1:
        i, bb = 0, b
        while bb != 1:
            bb = bb * bb % p
            i += 1
        g = pow(c, 1 << (m - i - 1), p)
        x, c, m = x * g % p, g * g % p, i
        b = b * c % p
    return x


G1_377_COFACTOR = (X377 - 1) ** 2 // 3


def ark_kzg10_setup_points():
    """(beta, g, gamma_g, h) exactly in the order ark-poly-commit 0.3.0 KZG10::setup draws them from ark_std::test_rng() [RECALL]"""
    rng = StdRngModel()
    q = Q377
    beta = rng.rand_fp(R377, 4)

    def g1():
        while True:
            x = rng.rand_fp(q, 6)
            greatest = rng.rand_bool()
            y = fp_sqrt((x * x * x + 1) % q, q)
            if y is None:
                continue
            ny = (q - y) % q
            y = y if (y < ny) != greatest else ny
            return ec_mul(G1_377_COFACTOR, (x, y), q)
    g = g1()
    gamma_g = g1()
    bt, _, cof = derive_g2_377()
    F = Fq2(q, -5)
    while True:
        x = (rng.rand_fp(q, 6), rng.rand_fp(q, 6))
        greatest = rng.rand_bool()
        y = F.sqrt(F.add(F.mul(F.mul(x, x), x), bt))
        if y is None:
            continue
        ny = ((q - y[0]) % q, (q - y[1]) % q)
        lt = (y[1], y[0]) < (ny[1], ny[0])              # QuadExtField order: c1 first, then c0
        y = y if lt != greatest else ny
        h = ec2_mul(F, cof, (x, y))
        break
    return beta, g, gamma_g, h


# ---------------- twisted Edwards model of BLS12-377 G1 (MSM kernels, csrc/te28.cuh) -------------------------------------------------
# y^2 = x^3 + 1 has the 2-torsion point (-1, 0) and 3 is a square mod q, so the curve has a Montgomery model B v^2 = u^3 + A u^2 + u
# (u, v) = (s (x + 1), s y), s = 1 / sqrt(3), A = -3 s, B = s, hence the twisted Edwards model a' X^2 + Y^2 = 1 + d' X^2 Y^2 with a' = (A + 2) / B,
# d' = (A - 2) / B, (X, Y) = (u / v, (u - 1) / (u + 1)); -a' is a square, so X -> f X with f = sqrt(-a') gives a = -1:
#       -x^2 + y^2 = 1 + d x^2 y^2,   d = -d' / a'.
# With a = -1 the unified extended-coordinate addition (Hisil-Wong-Carter-Dawson 2008) costs 7 field products when the second operand is a precomputed
# affine point (y - x, y + x, 2 d x y), against 10 for the XYZZ mixed addition on the Weierstrass model.  d is a square here, so the law is not
# complete on the whole curve; its exceptional cases need operands whose sum or difference has even order, which cannot happen inside the
# prime-order subgroup the KZG SRS lives in.
def edwards_377():
    q = Q377
    inv = lambda a: pow(a % q, -1, q)
    s3 = fp_sqrt(3, q)
    s3 = min(s3, q - s3)                              # canonical choice of the root
    s = inv(s3)
    A, B = (-3 * s) % q, s
    ap, dp = (A + 2) * inv(B) % q, (A - 2) * inv(B) % q
    f = fp_sqrt(-ap % q, q)
    f = min(f, q - f)
    d = (-dp * inv(ap)) % q
    return dict(s=s, si=s3, f=f, fi=inv(f), d=d, k2d=2 * d % q, sif=s3 * f % q)


def te_from_weierstrass(P, te=None):
    """affine Weierstrass point (or None) -> affine twisted Edwards point of the a = -1 model"""
    te = te or edwards_377()
    q = Q377
    if P is None:
        return (0, 1)
    x, y = P
    u, v = te["s"] * (x + 1) % q, te["s"] * y % q
    return (te["f"] * u * pow(v, -1, q) % q, (u - 1) * pow(u + 1, -1, q) % q)


def te_to_weierstrass(E, te=None):
    te = te or edwards_377()
    q = Q377
    xe, ye = E
    if xe == 0 and ye == 1:
        return None
    u = (1 + ye) * pow(1 - ye, -1, q) % q
    v = u * te["f"] * pow(xe, -1, q) % q
    return ((u * te["si"] - 1) % q, v * te["si"] % q)


def te_add(E1, E2, te=None):
    """unified affine addition on -x^2 + y^2 = 1 + d x^2 y^2"""
    te = te or edwards_377()
    q, d = Q377, te["d"]
    x1, y1 = E1
    x2, y2 = E2
    t = d * x1 * x2 * y1 * y2 % q
    return ((x1 * y2 + y1 * x2) * pow(1 + t, -1, q) % q, (y1 * y2 + x1 * x2) * pow(1 - t, -1, q) % q)


def _check_edwards_377():
    import random
    te = edwards_377()
    q, d = Q377, te["d"]
    rnd = random.Random(5)
    pts = [ec_mul(rnd.randrange(1, R377), G1_377, q) for _ in range(6)]
    for P in pts:
        xe, ye = te_from_weierstrass(P, te)
        assert (-xe * xe + ye * ye - 1 - d * xe * xe * ye * ye) % q == 0
        assert te_to_weierstrass((xe, ye), te) == P
    for P in pts[:3]:
        for Q in pts[3:]:
            for A, B in ((P, Q), (P, P), (P, None), (P, (P[0], (-P[1]) % q))):
                want = ec_add(A, B, q)
                got = te_add(te_from_weierstrass(A, te), te_from_weierstrass(B, te), te)
                assert te_to_weierstrass(got, te) == want


_check_edwards_377()
