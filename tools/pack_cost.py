"""Cost of the RGB888 packing copy that precedes the gather (distributed.BandGatherer.submit), alone on the GPU."""
import torch, time
dev = torch.device("cuda", 0)
n = 1920 * 1080
fb = torch.randint(0, 1 << 24, (n,), dtype=torch.int32, device=dev)
packed = torch.empty(n * 3, dtype=torch.uint8, device=dev)
bgr = fb.view(torch.uint8).view(-1, 4)[:, :3]
for _ in range(5):
    packed.view(-1, 3).copy_(bgr)
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(100):
    packed.view(-1, 3).copy_(bgr)
b.record()
torch.cuda.synchronize()
print("torch strided pack: %.4f ms per 1920x1080 frame" % (a.elapsed_time(b) / 100))
