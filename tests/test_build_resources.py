"""CPU: what the compiler made of the step kernels (hipcc cross-compiles gfx950 without a GPU).  The headline
instantiation of the streaming kernel must keep the occupancy its design counts on, and no instantiation may spill
to scratch."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_step_kernel_resources():
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import resource_usage
    rows = {r[0]: r for r in resource_usage.collect()}
    head = rows["void cpf::step_kernel_stream<false, true, false, false, false>"]
    vgpr, scratch, lds = int(head[1]), int(head[4]), int(head[8])
    assert scratch == 0 and vgpr <= 80 and lds <= 160 * 1024 // 24       # 6 waves per SIMD by registers and by LDS
    for name, r in rows.items():
        if "step_kernel_stream" in name or "step_kernel_coop" in name:
            assert int(r[4]) == 0 and int(r[7]) == 0, (name, r)            # no scratch, no VGPR spills
    assert len([n for n in rows if "step_kernel_stream" in n]) == 32
