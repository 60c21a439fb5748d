"""Generates the committed golden vectors by running the UNMODIFIED reference
(/root/reference, CPU, Python 3 + torch 2.x) through oracle/ref_harness.py.

    python tests/golden/make_golden.py

Only runs in the authoring container (the GPU box has no /root/reference).  Inputs that
the tests need at run time are committed next to the outputs: graf_img1.png /
graf_img6.png / graf_H1to6p are byte copies of the reference's test-graf data fixtures,
pretrained/AffNet.pth and pretrained/OriNet.pth are byte copies of the reference's
shipped checkpoints (data, not source).  HardNet++.pth is a missing blob in the
reference, so descriptors use oracle.synthetic_hardnet_state(0).
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import affnet_oracle as orc  # noqa: E402  (only for the synthetic inputs / synthetic HardNet weights)
import ref_harness as rh  # noqa: E402


def load_gray(path):
    from PIL import Image
    img = np.mean(np.array(Image.open(path).convert("RGB")), axis=2)  # hesaffnet.py:35-36
    return torch.from_numpy(img.astype(np.float32)).view(1, 1, img.shape[0], img.shape[1])


def pyramid_digest(pyr):
    """float64 sum and 16 strided samples per level: enough to pin every level bit-for-bit in
    practice without storing 20 MB."""
    sums, samples = [], []
    for octv in pyr:
        for lvl in octv:
            a = lvl[0, 0].numpy()
            sums.append(a.astype(np.float64).sum())
            flat = a.reshape(-1)
            samples.append(flat[np.linspace(0, flat.size - 1, 16).astype(np.int64)])
    return np.array(sums), np.stack(samples)


def main():
    ns = rh.import_reference()
    aff_sd, ori_sd = rh.load_state_dict("AffNet.pth"), rh.load_state_dict("OriNet.pth")
    hard_sd = orc.synthetic_hardnet_state(0)
    A = ns.architectures.AffNetFast(PS=32); A.load_state_dict(aff_sd); A.eval()
    O = ns.architectures.OriNetFast(PS=32); O.load_state_dict(ori_sd); O.eval()
    Hn = ns.HardNet.HardNet(); Hn.load_state_dict(hard_sd); Hn.eval()
    SSAPE = ns.SparseImgRepresenter.ScaleSpaceAffinePatchExtractor

    def full(x, n, do_ori=True):
        det = SSAPE(mrSize=5.192, num_features=n, border=5, num_Baum_iters=1, AffNet=A, OriNet=O)
        with torch.no_grad(), rh.quiet():
            L, r = det(x, do_ori=do_ori)
            P = det.extract_patches_from_pyr(L, PS=32)
            D = Hn(P)
        ps, pm = pyramid_digest(det.scale_pyr)
        return dict(LAFs=L.numpy(), resp=r.numpy(), desc=D.numpy(), patches_head=P[:16].numpy(),
                    patches_sum=P.numpy().astype(np.float64).sum(axis=(1, 2, 3)),
                    pyr_sums=ps, pyr_samples=pm, sigmas=np.array(det.sigmas), pix_dists=np.array(det.pix_dists))

    # 1. synthetic image, small: the CPU suite re-runs the oracle on this one
    x = orc.synthetic_image(240, 320, 1)
    gold = full(x, 300)
    # detector stage alone (SparseImgRepresenter.py:53-111 + :198): C = 450 candidates before AffNet
    det = SSAPE(mrSize=5.192, num_features=300, border=5, num_Baum_iters=1, AffNet=A, OriNet=O)
    with torch.no_grad(), rh.quiet():
        r_, L_, o_, l_ = det.multiScaleDetector(x, 450)
        L_[:, 0:2, 0:2] = 5.192 * L_[:, :, 0:2]
        Lpx = ns.LAF.denormalizeLAFs(L_, 320, 240)
    gold.update(det_resp=r_.numpy(), det_LAFs_px=Lpx.numpy(), det_oct=o_.numpy(), det_lev=l_.numpy())
    np.savez_compressed(os.path.join(HERE, "synth_240x320_s1_n300.npz"), **gold)
    # 2. config 2: test-graf/img1.png (N=500 keeps the fixture small; N=2000 is checked live vs the oracle)
    g1 = load_gray(os.path.join(HERE, "graf_img1.png"))
    np.savez_compressed(os.path.join(HERE, "graf_img1_n500.npz"), **full(g1, 500))
    # 3. no-orientation + threshold mode as shipped in hesaffnet.py (th=-1 -> num=-1) and Oxford ellipses
    det = SSAPE(mrSize=5.192, num_features=300, border=5, num_Baum_iters=1, th=-1, AffNet=A)
    with torch.no_grad(), rh.quiet():
        L, r = det(x)
    np.savez_compressed(os.path.join(HERE, "synth_240x320_s1_thmode.npz"), LAFs=L.numpy(), resp=r.numpy(),
                        ells=ns.LAF.LAFs2ell(L.numpy()))
    # 4. config 1: detect_affine_shape on a 32-px wide patch column cut from graf img1
    #    (face.png is a missing blob; examples/just_shape/detect_affine_shape.py:36-70)
    img_u8 = np.round(g1[0, 0].numpy()).astype(np.uint8)
    tiles = [img_u8[40 + 9 * i: 72 + 9 * i, 60 + 11 * i: 92 + 11 * i] for i in range(64)]
    col = np.concatenate(tiles, axis=0)
    patches = torch.from_numpy(col.reshape(64, 1, 32, 32).astype(np.float32) / 255.0)
    with torch.no_grad():
        out = A(patches).reshape(-1, 4).numpy()
    np.savez_compressed(os.path.join(HERE, "just_shape_column.npz"), column=col, affine=out)
    # 5. CNN-only vectors on random patches (AffNet / OriNet real weights, HardNet synthetic)
    g = torch.Generator().manual_seed(7)
    rp = torch.rand(48, 1, 32, 32, generator=g) * 255.0
    with torch.no_grad():
        np.savez_compressed(os.path.join(HERE, "cnn_random_patches.npz"), patches=rp.numpy(),
                            affnet=A(rp).numpy(), orinet=O(rp).numpy(), hardnet=Hn(rp).numpy())
    # 6. stand-alone sampler vectors: LAF.extract_patches on a pyramid-like image incl. out-of-image LAFs
    g = torch.Generator().manual_seed(11)
    n = 40
    lafs = torch.zeros(n, 2, 3)
    ang = torch.rand(n, generator=g) * 6.28
    sc = torch.rand(n, generator=g) * 0.2 + 0.02
    lafs[:, 0, 0] = sc * torch.cos(ang); lafs[:, 0, 1] = sc * torch.sin(ang) * 0.7
    lafs[:, 1, 0] = -sc * torch.sin(ang); lafs[:, 1, 1] = sc * torch.cos(ang) * 1.3
    lafs[:, :, 2] = torch.rand(n, 2, generator=g) * 1.2 - 0.1
    with torch.no_grad():
        p32 = ns.LAF.extract_patches(x, lafs, PS=32)
        p41 = ns.LAF.extract_patches(x, lafs, PS=41)
    np.savez_compressed(os.path.join(HERE, "sampler_synth.npz"), lafs=lafs.numpy(), p32=p32.numpy(), p41=p41.numpy())
    print("golden vectors written to", HERE)


if __name__ == "__main__":
    main()
