"""A small copy beside a big one: the DMA engines take transfers in submission order whatever the stream (why the streaming
path moves its small transfers with a kernel, k_copy_bytes)."""
import torch, time
big_h = torch.empty(150 << 20, dtype=torch.uint8).pin_memory()
big_d = torch.empty(150 << 20, dtype=torch.uint8, device='cuda')
small_d = torch.zeros(128 << 10, dtype=torch.uint8, device='cuda')
small_h = torch.empty(128 << 10, dtype=torch.uint8).pin_memory()
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
torch.cuda.synchronize()
for rep in range(3):
    with torch.cuda.stream(s1):
        big_d.copy_(big_h, non_blocking=True)
    t0 = time.perf_counter()
    with torch.cuda.stream(s2):
        small_h.copy_(small_d, non_blocking=True)
    s2.synchronize()
    t1 = time.perf_counter()
    s1.synchronize()
    t2 = time.perf_counter()
    print('small D2H beside a 150 MB H2D: %.3f ms; the H2D: %.3f ms' % ((t1 - t0) * 1e3, (t2 - t0) * 1e3))
for rep in range(2):
    t0 = time.perf_counter()
    with torch.cuda.stream(s2):
        small_h.copy_(small_d, non_blocking=True)
    s2.synchronize()
    print('small D2H alone: %.3f ms' % ((time.perf_counter() - t0) * 1e3))
# small H2D beside big H2D
small_h2 = torch.empty(128 << 10, dtype=torch.uint8).pin_memory()
for rep in range(2):
    with torch.cuda.stream(s1):
        big_d.copy_(big_h, non_blocking=True)
    t0 = time.perf_counter()
    with torch.cuda.stream(s2):
        small_d.copy_(small_h2, non_blocking=True)
    s2.synchronize()
    t1 = time.perf_counter()
    s1.synchronize()
    print('small H2D beside a 150 MB H2D: %.3f ms' % ((t1 - t0) * 1e3))
