"""Host-side access to raz-rng-v1 (Philox4x32-10 keyed by (seed, game id); csrc/raz_detmath.h) for the
two per-game draws the host needs (worker/self_play.py:144,182).  Pure-Python integer arithmetic —
a 10-round block cipher on four words, called once per finished game."""

M0, M1, W0, W1, MASK = 0xD2511F53, 0xCD9E8D57, 0x9E3779B9, 0xBB67AE85, 0xFFFFFFFF


def philox4x32_10(ctr, key):
    c0, c1, c2, c3 = ctr
    k0, k1 = key
    for _ in range(10):
        p0, p1 = M0 * c0, M1 * c2
        c0, c1, c2, c3 = ((p1 >> 32) ^ c1 ^ k0) & MASK, p1 & MASK, ((p0 >> 32) ^ c3 ^ k1) & MASK, p0 & MASK
        k0, k1 = (k0 + W0) & MASK, (k1 + W1) & MASK
    return c0, c1, c2, c3


def rng_pair(seed, game, purpose, event, sub=0, idx=0):
    r = philox4x32_10((idx, sub, event, purpose), (seed & MASK, game & MASK))
    u = lambda a, b: ((a >> 5) * 67108864.0 + (b >> 6)) / 9007199254740992.0
    return u(r[0], r[1]), u(r[2], r[3])
