#!/usr/bin/env python
"""Bank model of the split-operand trunks' LDS layouts (affnet_amd/csrc/cnn_mfma.h: LayQ = three bf16 terms in 48-byte cells, LayR = two fp16 terms,
16-byte pixels): LDS-array cycles of the fragment reads (ds_read_b128) of every loop and of the epilogue stores (ds_write_b64), from the access rules of
/opt/skills/guides/MI355X_MICROARCH.md, section LDS:
    ds_read_b128 : four service groups of 16 lanes - {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}, {32-35, 44-47, 52-59}, {36-43, 48-51, 60-63} -, 64 banks of
                   4 bytes (bank = (address / 4) mod 64), one cycle per group when no bank is asked twice, else the largest multiplicity  -> 4 cycles conflict free
    ds_write_b64 : four groups of 16 contiguous lanes, 32 banks (bank = (address / 4) mod 32)                                             -> 4 cycles conflict free
A reader's lane addresses are those of conv3x3_mfma_s3q (lane = pixel m = lane & 15 of a 16-pixel tile, lane quarter kq = lane >> 4 = its 8-channel group);
tile, tap and term offsets are the same for all lanes of an instruction and drop out.  tests/test_host_mirror.py pins the figures the design relies on.

    python tools/lds_bank_model.py        -> the table (profiles/r04_lds_bank_model.txt is this output)
"""
import collections

READ_GROUPS = [list(range(0, 4)) + list(range(12, 16)) + list(range(20, 28)), list(range(4, 12)) + list(range(16, 20)) + list(range(28, 32)),
               list(range(32, 36)) + list(range(44, 48)) + list(range(52, 60)), list(range(36, 44)) + list(range(48, 52)) + list(range(60, 64))]
WRITE_GROUPS = [list(range(16 * g, 16 * g + 16)) for g in range(4)]


class LayQ:
    """48-byte cells, the terms of a pixel side by side (cnn_mfma.h)."""
    def __init__(self, H, W, WP, C, GREM=0):
        self.H, self.W, self.WP, self.C = H, W, WP, C
        self.PIXB, self.ROWB, self.TSTEP = 48, WP * 48, 16
        self.GS = ((H + 2) * self.ROWB + 255) // 256 * 256 + GREM

    def at(self, y, x):
        return y * self.ROWB + x * self.PIXB


class LayR(LayQ):
    """16-byte pixels, the hi cells and the lo cells of a row side by side (two terms)."""
    def __init__(self, H, W, WP, C, GREM=0, HALLOC=None):
        self.H, self.W, self.WP, self.C = H, W, WP, C
        self.PIXB, self.ROWB, self.TSTEP = 16, 2 * WP * 16, WP * 16
        self.GS = ((HALLOC or H + 2) * self.ROWB + 255) // 256 * 256 + GREM


def cycles(addrs, groups, nbanks, width):
    """LDS-array cycles of one wave instruction: per service group the largest number of lanes on one bank."""
    total = 0
    for grp in groups:
        use = collections.Counter()
        for lane in grp:
            for b in range(width // 4):
                use[(addrs[lane] // 4 + b) % nbanks] += 1
        total += max(use.values())
    return total


def read_cycles(L, stride, c16=False):
    """ds_read_b128 of one fragment of a loop over layout L (input) at the given stride."""
    wout = L.W // stride
    addrs = []
    for lane in range(64):
        m, kq = lane & 15, lane >> 4
        oy, ox = m // wout, m % wout
        addrs.append(((kq & 1) if c16 else kq) * L.GS + L.at(oy * stride, ox * stride))
    return cycles(addrs, READ_GROUPS, 64, 16)


def write_cycles(L):
    """ds_write_b64 of the epilogue (split_store4): lane = pixel n = lane & 15 of a tile, quad g = lane >> 4 of a 16-channel tile."""
    addrs = []
    for lane in range(64):
        n, g = lane & 15, lane >> 4
        oy, ox = n // L.W, n % L.W
        addrs.append((g >> 1) * L.GS + L.at(oy + 1, ox + 1) + 8 * (g & 1))
    return cycles(addrs, WRITE_GROUPS, 32, 8)


def readers():
    """(name, layout, stride, 16-input-channel loop) of every split-operand loop of the trunks (cnn32.hip) - HardNet (CB = 32) and AffNet / OriNet (CB = 16)."""
    out = []
    for cb, net in ((32, "HardNet"), (16, "AffNet / OriNet")):
        c16 = cb == 16
        # three bf16 terms: LayQ, conv0 .. conv2 on half patches
        out += [("%s fp32_split3 conv1 (stride 1, 32-wide)" % net, LayQ(16, 32, 34, cb, 0), 1, c16),
                ("%s fp32_split3 conv2 (stride 2, 32-wide)" % net, LayQ(16, 32, 34, cb, 16), 2, c16),
                ("%s fp32_split3 conv3 (stride 1, 16-wide)" % net, LayQ(16, 16, 18, 2 * cb, 0), 1, False),
                ("%s fp32_split3 conv4 (stride 2, 16-wide)" % net, LayQ(16, 16, 18, 2 * cb, 0), 2, False),
                ("%s fp32_split3 conv5 (stride 1, 8-wide)" % net, LayQ(8, 8, 16, 4 * cb, 128), 1, False)]
        # two fp16 terms: LayR, whole patch
        g2 = 16 if cb == 32 else 0            # conv2 / conv3 outputs: HardNet takes conv4's group stride, the 16-channel nets conv3's
        out += [("%s fp32_split2h conv1 (stride 1, 32-wide)" % net, LayR(32, 32, 34, cb, 0), 1, c16),
                ("%s fp32_split2h conv2 (stride 2, 32-wide)" % net, LayR(32, 32, 34, cb, 16), 2, c16),
                ("%s fp32_split2h conv3 (stride 1, 16-wide)" % net, LayR(16, 16, 20, 2 * cb, g2), 1, False),
                ("%s fp32_split2h conv4 (stride 2, 16-wide)" % net, LayR(16, 16, 20, 2 * cb, g2), 2, False),
                ("%s fp32_split2h conv5 (stride 1, 8-wide)" % net, LayR(8, 8, 12, 4 * cb, 0), 1, False)]
    return out


def main():
    print("%-58s %-34s %s" % ("loop (reader of its input layout)", "layout: row pitch B, GS mod 256", "LDS cycles per ds_read_b128 (4 = conflict free)"))
    for name, L, stride, c16 in readers():
        print("%-58s %-34s %d" % (name, "%s %5d B, %3d" % (type(L).__name__, L.ROWB, L.GS % 256), read_cycles(L, stride, c16)))
    print()
    print("epilogue stores (ds_write_b64, one 8-byte half of a 16-byte cell slot per lane; 4 = conflict free):")
    for name, L in (("LayQ 32-wide", LayQ(16, 32, 34, 32)), ("LayQ 16-wide", LayQ(16, 16, 18, 64)), ("LayQ 8-wide", LayQ(8, 8, 16, 128, 128)),
                    ("LayR 32-wide", LayR(32, 32, 34, 32)), ("LayR 16-wide", LayR(16, 16, 20, 64, 16)), ("LayR 8-wide", LayR(8, 8, 12, 128))):
        print("  %-14s %d" % (name, write_cycles(L)))


if __name__ == "__main__":
    main()
