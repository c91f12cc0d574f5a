import ctypes as C, torch, os
here = os.path.dirname(os.path.abspath(__file__))
L = C.CDLL(os.path.join(here, "store_bw.so"))
L.store_bw.argtypes = [C.c_void_p, C.c_longlong] + [C.c_int] * 5 + [C.c_void_p]
npix = 128 * 240 * 320
dst = torch.zeros(npix * 128, device="cuda")
st = torch.cuda.current_stream().cuda_stream
for grid in (256, 512, 2048):
    for spin in (0, 200):
        for shape in (0, 1):
            for _ in range(3): L.store_bw(dst.data_ptr(), npix // 32, 512, 128, shape, spin, grid, st)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); n = 20
            for _ in range(n): L.store_bw(dst.data_ptr(), npix // 32, 512, 128, shape, spin, grid, st)
            e1.record(); torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / n
            print("grid %5d (x4 waves) spin %3d shape %d (%s): %7.1f us  %5.2f TB/s" % (grid, spin, shape, "16 x dword, lane = channel" if shape == 0 else "4 x 16 B, lane = pixel", us, npix * 128 / us / 1e6), flush=True)
