"""Practical HBM roof of the box (SURVEY.md 8d asks for it beside the 8 TB/s nominal peak): device-to-device copy
of a buffer far larger than the 256 MB Infinity Cache, HIP-event timed. Bytes moved = 2 x size (read + write)."""
import torch, json
n = 4 << 30
a = torch.empty(n, dtype=torch.uint8, device="cuda"); b = torch.empty_like(a); a.fill_(1)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
b.copy_(a); torch.cuda.synchronize()
best = 1e9
for _ in range(5):
    e0.record(); b.copy_(a); e1.record(); torch.cuda.synchronize(); best = min(best, e0.elapsed_time(e1))
print(json.dumps({"copy_GiB": n / 2**30, "ms": round(best, 3), "read_plus_write_GB_per_s": round(2 * n / best / 1e6, 1), "device": torch.cuda.get_device_name(0)}))
