"""Host issue time vs GPU time of one leaf_cnn training step (MI355X only)."""
import sys, time, math
import torch
sys.path.insert(0, ".")
from leaffliction_amd import _lib
from leaffliction_amd.model.cnn import LeafCNN

_lib.load()
dev = torch.device("cuda", 0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
m = LeafCNN(num_classes=8, img_size=224, widths=(32, 64, 128, 256), drop_block=0.15, drop_top=0.40,
            l2_reg=1e-4, augment=True, use_se=True, seed=42, device=dev)
x = torch.randint(0, 256, (n, 224, 224, 3), dtype=torch.uint8).to(dev)
y = torch.nn.functional.one_hot(torch.randint(0, 8, (n,)), 8).float().to(dev)
for _ in range(3):
    m.train_step(x, y, 1e-3)
torch.cuda.synchronize()
for tag in ("a", "b", "c", "d", "e", "f"):
    host = []
    t0 = time.perf_counter()
    for _ in range(10):
        h0 = time.perf_counter()
        m.train_step(x, y, 1e-3)
        host.append(time.perf_counter() - h0)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"host issue/step {1e3*sum(host)/10:.2f} ms (min {1e3*min(host):.2f})  total/step {1e3*(t2-t0)/10:.2f} ms  tail sync {1e3*(t2-t1):.2f} ms")
# host-only cost of the pieces
t0 = time.perf_counter()
for _ in range(10):
    m.draw_dropout(n)
torch.cuda.synchronize()
print(f"draw_dropout {1e2*(time.perf_counter()-t0):.2f} ms")
t0 = time.perf_counter()
for _ in range(10):
    m.draw_augmentation(n)
torch.cuda.synchronize()
print(f"draw_augmentation {1e2*(time.perf_counter()-t0):.2f} ms")
import cProfile, pstats
pr = cProfile.Profile()
pr.enable()
for _ in range(5):
    m.train_step(x, y, 1e-3)
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(25)
