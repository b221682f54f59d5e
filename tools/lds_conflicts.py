"""Bank-conflict model of the marching streaming kernel's LDS plane layout (csrc/mconv.hip), checked exhaustively on the CPU.

A plane row holds TZ voxels x G 16-byte pieces (G = CIN/8) in the order [piece'][z], piece' = (piece + 2 * (row * RS / 16)) % G with
RS = TZ * G slots per row.  An MFMA B-operand read is one ds_read_b128 per lane: lane (g, l15) reads K-group p = ks*4 + g -> (tap p / G, piece
p % G) of voxel (row = 1 + mt*(16/TZ) + l15 / TZ + dy(tap), z = l15 % TZ).  ds_read_b128 is serviced in four 16-lane groups
({0-3,12-15,20-27}, {4-11,16-19,28-31}, +32; MI355X_MICROARCH.md §LDS); a group is conflict-free iff its 16 slots are distinct mod 16.
"""
import itertools

GROUPS = [[0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27], [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31]]
GROUPS += [[l + 32 for l in g] for g in GROUPS]


def slot(row, z, piece, tz, g):
    rs = tz * g
    return row * rs + ((piece + 2 * (row * rs // 16)) % g) * tz + z


def extra_cycles(cin, tz, taps=9, mt_rows=None):
    """Extra LDS cycles (bank conflicts) of all operand reads of one M-tile, and the conflict-free cycle count."""
    g = cin // 8
    ksteps = (taps * g + 3) // 4
    extra = base = 0
    for mt in range(2):  # M-tile parity cannot matter (16 / TZ rows per tile), checked anyway
        for ks in range(ksteps):
            for grp in GROUPS:
                seen = {}
                for lane in grp:
                    gg, l15 = lane >> 4, lane & 15
                    p = ks * 4 + gg
                    tap, piece = p // g, p % g
                    if tap >= taps:  # padded K-group (zero weights): the kernel reads K-group p - taps*G, a genuine tap of the same voxel
                        tap, piece = (p - taps * g) // g, (p - taps * g) % g
                    dy = tap % 3 - 1 if taps == 9 else 0
                    row = 1 + mt * (16 // tz) + l15 // tz + dy
                    s = slot(row, l15 % tz, piece, tz, g) + (tap // 3) * 16 * 1000  # planes are 256-byte aligned
                    seen.setdefault(s % 16, set()).add(s)
                ways = max(len(v) for v in seen.values())
                extra += ways - 1
                base += 1
    return extra, base


if __name__ == "__main__":
    for cin, tz in itertools.product((8, 16, 32, 64), (2, 4, 8)):
        if 16 % tz:
            continue
        e, b = extra_cycles(cin, tz)
        print(f"CIN {cin:3d} TZ {tz}: {e} extra cycles on {b} ({100.0 * e / b:.0f} %)")
