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
    head = rows["void cpf::step_kernel_stream<false, true, false, false, 0>"]
    vgpr, sgpr, scratch, lds = int(head[1]), int(head[3]), int(head[4]), int(head[8])
    # 7 waves per SIMD = 28 single-wave workgroups per CU, by ALL three resources (MI355X_MICROARCH.md): <= 72 VGPRs
    # (512 / 7, granule 8), <= 96 SGPRs (800 per SIMD: floor(800 / (alloc + 16)) with a granule of 16; the compiler's own
    # occupancy figure does not know this limit) and <= 160 KB / 28 of LDS
    assert scratch == 0 and vgpr <= 72 and sgpr <= 96 and lds <= 160 * 1024 // 28, head
    big = rows["void cpf::step_kernel_stream<false, true, false, false, 1>"]           # large meshes: the same
    assert int(big[1]) <= 72 and int(big[3]) <= 96 and int(big[8]) <= 160 * 1024 // 28, big
    brown = rows["void cpf::step_kernel_stream<true, true, false, false, 0>"]         # tutorial diffusion: 6 waves
    assert int(brown[1]) <= 80 and int(brown[8]) <= 160 * 1024 // 24, brown
    flat = rows["void cpf::step_kernel_stream<false, true, false, false, 8>"]           # what the headline configuration runs
    assert int(flat[1]) <= 72 and int(flat[3]) <= 96 and int(flat[8]) <= 160 * 1024 // 28, flat
    for name, r in rows.items():
        if "step_kernel_stream" in name or "step_kernel_coop" in name:
            assert int(r[4]) == 0 and int(r[7]) == 0, (name, r)            # no scratch, no VGPR spills
    assert len([n for n in rows if "step_kernel_stream<" in n]) == 144
    # the "VertexVelocity" cycle on the streaming kernel (round 6): loop and fixed lookup only; the tet tables of its advect cost
    # it half the occupancy (<= 128 VGPRs: 4 waves), not a byte of scratch
    vertex = {n: r for n, r in rows.items() if "step_kernel_stream_vertex<" in n}
    assert len(vertex) == 32 and all(n.endswith((", 0>", ", 1>")) for n in vertex)
    v0 = vertex["void cpf::step_kernel_stream_vertex<false, true, false, false, 0>"]
    assert int(v0[1]) <= 128 and int(v0[4]) == 0 and int(v0[8]) <= 160 * 1024 // 16, v0       # (LDS: + three cells' cone rows)
