import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import affnet_amd
from affnet_amd import ReprojectionStuff as RS, Losses
gen = torch.Generator().manual_seed(5)
for n in (3000, 2048, 1024, 640, 100):
    x = torch.nn.functional.normalize(torch.randn(n, 128, generator=gen), dim=1).cuda()
    t1, t2, md, md2 = RS.match_snn(x, x, 0.8)
    bad = torch.nonzero(md > 0.01).flatten()
    full = Losses.distance_matrix_vector(x, x)
    dg = full.diagonal()
    print(n, "bad rows", bad.numel(), bad[:20].tolist(), "diag max", float(dg.max()), "nan", int(torch.isnan(full).sum()))
    if bad.numel():
        i = int(bad[0]); print(" row", i, "md", float(md[i]), "full row min", float(torch.nan_to_num(full[i], nan=9).min()), "diag", float(full[i, i]))
