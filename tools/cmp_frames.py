import sys, torch
a, b = (torch.load(f"gpurun_out/frame_{t}.pt") for t in sys.argv[1:3])
for k in ("image", "depth", "weights_sum"):
    x, y = a[k].float().nan_to_num(), b[k].float().nan_to_num()
    print(k, "equal" if torch.equal(x, y) else "DIFFERENT: %d of %d elements, max abs %.3g" % (int((x != y).sum()), x.numel(), (x - y).abs().max().item()))
print("stats", a["stats"].tolist(), b["stats"].tolist())
