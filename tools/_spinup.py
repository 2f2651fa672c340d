"""Developer tools (GPU box): bring the device to its steady clocks before a measurement.  An MI355X that has been idle
needs ~30 ms of load to get there (tools/drift_check.py: 0.148 -> 0.119 ms per launch of the same work); a measurement of
20 launches right after the set-up would be a measurement of the ramp."""
import time


def device_spinup(ctx, torch, x, y, z, cell, n, dt, ms=100.0):
    """Step launches on SCRATCH copies of the particle arrays (statistics-on instantiation, so that a profiler's average
    of the statistics-off one covers the measured launches only); the caller's arrays are untouched."""
    sx, sy, sz, sc = x[:n].clone(), y[:n].clone(), z[:n].clone(), cell[:n].clone()
    ctx.set_option("stats", 1)
    torch.cuda.synchronize()
    t0, launches = time.perf_counter(), 0
    while (time.perf_counter() - t0) * 1e3 < ms:
        for _ in range(20):
            ctx.step_dev(sx.data_ptr(), sy.data_ptr(), sz.data_ptr(), sc.data_ptr(), None, None, n, dt, 0.0, 0, 1, 0)
        launches += 20
        torch.cuda.synchronize()
    ctx.set_option("stats", 0)
    return launches
