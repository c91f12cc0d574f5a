import ctypes as C, torch, subprocess, os, sys
here = os.path.dirname(os.path.abspath(__file__))
L = C.CDLL(os.path.join(here, "slice_bw.so"))
L.slice_copy.argtypes = [C.c_void_p, C.c_void_p, C.c_longlong] + [C.c_int] * 7 + [C.c_void_p]
npix = 128 * 240 * 320
src = torch.randn(npix * 128, device="cuda"); dst = torch.zeros(npix * 128, device="cuda")
st = torch.cuda.current_stream().cuda_stream
def run(name, sp, so, dp, dof, sb, mode, grid=2048):
    for _ in range(3): L.slice_copy(src.data_ptr(), dst.data_ptr(), npix, sp, so, dp, dof, sb, mode, grid, st)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    n = 20
    for _ in range(n): L.slice_copy(src.data_ptr(), dst.data_ptr(), npix, sp, so, dp, dof, sb, mode, grid, st)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / n
    by = npix * sb * ((mode & 1) + ((mode >> 1) & 1))
    print("%-58s %7.1f us  %5.2f TB/s useful" % (name, us, by / us / 1e6), flush=True)
for grid in (1024, 2048, 8192):
    print("grid", grid)
    run("copy contiguous 128 B/px -> contiguous", 128, 0, 128, 0, 128, 3, grid)
    run("read only contiguous", 128, 0, 128, 0, 128, 1, grid)
    run("write only contiguous", 128, 0, 128, 0, 128, 2, grid)
    run("read slice [128,256) of 512-B pixels only", 512, 128, 128, 0, 128, 1, grid)
    run("write slice [0,128) of 512-B pixels only", 128, 0, 512, 0, 128, 2, grid)
    run("slice [128,256) of 512 -> slice [0,128) of 512 (other buffer)", 512, 128, 512, 0, 128, 3, grid)
    run("slice [128,256) of 512 -> slice [256,384) of 512 (other buffer)", 512, 128, 512, 256, 128, 3, grid)
    run("slices [128,384) of 512 (256 B) -> slice [384,512)", 512, 128, 512, 384, 256, 1, grid)
    run("contiguous 256 B/px read only", 256, 0, 128, 0, 256, 1, grid)
    run("whole 512-B pixels read only", 512, 0, 512, 0, 512, 1, grid)
