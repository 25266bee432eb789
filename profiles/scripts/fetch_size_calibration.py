"""Calibrate rocprofv3's FETCH_SIZE / WRITE_SIZE for THIS access pattern: random 16-byte gathers from a table far larger
than the Infinity Cache, and a streaming copy.  Run under
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d <dir> -- python fetch_size_calibration.py
(and again with WRITE_SIZE); the script prints the byte counts the kernels must move, the CSV has the counters."""
import torch
torch.manual_seed(0)
dev = torch.device("cuda", 0)
N = 1 << 27  # 128 M rows x 16 B = 2 GiB table
table = torch.empty((N, 4), dtype=torch.int32, device=dev)
idx = torch.randint(0, N, (1 << 24,), device=dev)  # 16 M random rows
out = torch.empty((idx.numel(), 4), dtype=torch.int32, device=dev)
src = torch.empty(1 << 30, dtype=torch.uint8, device=dev)
dst = torch.empty_like(src)
torch.cuda.synchronize()
for _ in range(3):
    torch.index_select(table, 0, idx, out=out)   # gather: 16 M x 16 B useful = 268 MB; 16 M x 64 B sectors = 1074 MB
    dst.copy_(src)                                # stream: 1 GiB read + 1 GiB written
torch.cuda.synchronize()
print("gather: rows %d, useful bytes %d, 64-byte sectors %d bytes, index bytes %d, output bytes %d" %
      (idx.numel(), idx.numel() * 16, idx.numel() * 64, idx.numel() * 8, idx.numel() * 16))
print("copy: %d bytes read, %d bytes written" % (src.numel(), src.numel()))
