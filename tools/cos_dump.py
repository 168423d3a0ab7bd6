import sys, os
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")
import torch
import test_zz_bf16_cosine_gpu as T
gpu = torch.device("cuda:0")
gg32, gd32 = T._s2_grads(gpu, torch.float32)
gg16, gd16 = T._s2_grads(gpu, torch.bfloat16)
for what, a, b in (("G", gg32, gg16), ("D", gd32, gd16)):
    cs = T._compare(a, b, 1e-4, what)
    print(what, len(cs), "below 0.995:")
    for c, k in cs:
        if c < 0.995:
            print(f"   {c:.4f} {k} rms32={float(a[k].double().pow(2).mean().sqrt()):.3e} shape={tuple(a[k].shape)}")
g32 = T._s1_grads(gpu, torch.float32); g16 = T._s1_grads(gpu, torch.bfloat16)
cs = T._compare(g32, g16, 1e-4, "s1")
print("s1", len(cs), "min", cs[0], "below 0.993:", [(round(c,4),k) for c,k in cs if c < 0.993][:40])
import statistics
print("s1 median", statistics.median(c for c,_ in cs))
