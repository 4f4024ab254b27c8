"""JPEG.Table.Quantization.z(k:h:) -- decode.swift:1289-1298, closed form."""


def z(k: int, h: int) -> int:
    p = 1 if k + h < 8 else 0
    q = (k + h) & 1
    a = 72 * (p ^ 1)
    b = 2 * p - 1
    n = b * (k + h) - 14 * p + 15
    t = (n * (n + 1)) >> 1
    return a + b * t - q * k - (q ^ 1) * h - 1


# ZIGZAG[h][k]
ZIGZAG = [[z(k, h) for k in range(8)] for h in range(8)]
