"""PCIe-inclusive rate of the default workload: what a caller pays when frames come from host memory and
results go back to it.  Per frame: 640x480 gray u8 + 640x480 depth u16 up, ~80 KB of results (keypoints, descriptors,
keylines, matches) down, pinned host buffers, one stream.  Prints the copy rates and the frames/s implied
for copies and compute in series and overlapped.  usage: python tools/pcie_rate.py [frames] [compute_ms_per_batch]"""
import sys
import time

import torch


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 6144
    compute_ms = float(sys.argv[2]) if len(sys.argv) > 2 else 112.8
    dev = torch.device("cuda:0")
    up = 640 * 480 * (1 + 2)
    down = 80 * 1024
    chunk = 512  # frames per copy call
    h_up = torch.empty((chunk, up), dtype=torch.uint8).pin_memory()
    h_dn = torch.empty((chunk, down), dtype=torch.uint8).pin_memory()
    d_up = torch.empty((chunk, up), dtype=torch.uint8, device=dev)
    d_dn = torch.empty((chunk, down), dtype=torch.uint8, device=dev)
    for _ in range(2):
        d_up.copy_(h_up, non_blocking=True); h_dn.copy_(d_dn, non_blocking=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n // chunk):
        d_up.copy_(h_up, non_blocking=True)
    torch.cuda.synchronize()
    t_up = time.perf_counter() - t0
    t0 = time.perf_counter()
    for _ in range(n // chunk):
        h_dn.copy_(d_dn, non_blocking=True)
    torch.cuda.synchronize()
    t_dn = time.perf_counter() - t0
    gb_up, gb_dn = n * up / 1e9, n * down / 1e9
    print(f"H2D {gb_up:.2f} GB in {t_up * 1e3:.1f} ms = {gb_up / t_up:.1f} GB/s; D2H {gb_dn:.2f} GB in {t_dn * 1e3:.1f} ms = {gb_dn / t_dn:.1f} GB/s")
    ser = n / (t_up + t_dn + compute_ms / 1e3)
    ovl = n / max(t_up, t_dn, compute_ms / 1e3)
    print(f"{n} frames, compute {compute_ms} ms: in series {ser:.0f} frames/s, copies overlapped with compute {ovl:.0f} frames/s")


if __name__ == "__main__":
    main()
