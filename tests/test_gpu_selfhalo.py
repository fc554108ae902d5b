"""GPU: self-halo (dflo_hip_multi_create_self) -- ONE part that is its own neighbour across a virtual cut, driven through the
complete stage schedule of a multi-device run on one GPU: rim shards on the comm stream beside the interior on the compute
stream, pack, transport into the trace table (grouped ncclSend / ncclRecv to itself on a one-rank RCCL communicator, or the
delivering pack kernels, or staging + copy, or the sequence-word transport of DFLO_RANK_TRANSPORT=ipc against itself), the averages that feed the LxF flux / the TVB limiter of the rim, and the
time-step reduction (ncclAllReduce(min) over the one rank).  It is what bench.py --self-halo times; here: the results are
those of the single engine -- bit for bit on the nodal basis.

What the schedule replaces: update_ghost_values (src_mpi/claw.cc:793, src_mpi/limiter.cc:232), Utilities::MPI::min
(src_mpi/claw.cc:579)."""
import numpy as np
import pytest

import dflo_amd
import test_gpu_multi as S          # the small configurations (every shard is rim)
import test_gpu_multi_large as L    # 512^2 C2 and the 1001 x 1000 slab pair of C4 (real interior shards)

pytestmark = pytest.mark.gpu


def _self(mesh, prm, transport, method="slab"):
    claw = dflo_amd.MultiConservationLaw.for_self(mesh, prm, 0, transport=transport, partitioner=method)
    assert claw.n_parts == 1 and claw.n_local == 1
    own, ghost = claw.part_cells(0)
    assert (own == np.arange(mesh.n_cells)).all() and len(ghost) > 0      # owns everything, and still has ghost cells
    return claw


@pytest.mark.parametrize("transport", ["rccl", "direct", "copy", "ipc"])
def test_c2_512_self_halo_bit_identical_to_the_single_engine(transport):
    ref = L.reference("c2")
    mesh, prm, ic, programs = L.case("c2")
    claw = _self(mesh, prm, transport)
    L._plan_has_interior(claw)
    cnt, rk, what = claw.comm_info()
    assert "self-halo" in what and "periodic seam" in what
    if transport == "rccl":
        assert (cnt, rk) == (1, 0) and "ncclSend" in what      # as the communicator itself reports them
    # one rank of the weak-scaling run: 2 x 512 cut faces seen from this side, the same number of ghost cells
    assert len(claw.part_cells(0)[1]) == 2 * 512
    L.setup(claw, mesh, ic, programs)
    claw.exchange_timing(True)
    got = L.run(claw, False)
    us, n = claw.exchange_timing(False)
    claw.close()
    assert n > 0 and us > 0.0                                    # the exchanges happened (and were timed on the comm stream)
    assert got["dt"] == ref["dt"] and got["t"] == ref["t"]
    assert np.array_equal(got["avg"], ref["avg"])
    assert np.array_equal(got["u"], ref["u"])


@pytest.mark.parametrize("transport", ["rccl", "direct", "ipc"])
def test_c4_slab_self_halo_matches_the_single_engine(transport):
    """TVB + positivity + moving inflow: over RCCL one exchange per stage (the cut cells unlimited with their neighbours' averages,
    the ghost cells limited by the receiver), with the delivering pack kernels two (averages before the rim limiter, traces after it),
    over the IPC transport both inside the stage kernel and the limiter pass"""
    ref = L.reference("c4")
    mesh, prm, ic, programs = L.case("c4")
    claw = _self(mesh, prm, transport)
    assert "cut through the middle" in claw.comm_info()[2]
    L._plan_has_interior(claw)
    L.setup(claw, mesh, ic, programs)
    got = L.run(claw, True)
    claw.close()
    assert got["dt"] == ref["dt"] and got["t"] == ref["t"]
    assert L.rel(got["avg"], ref["avg"]) < 1e-9
    assert L.rel(got["u"], ref["u"]) < 1e-8


@pytest.mark.parametrize("config", ["c4", "c3"])
@pytest.mark.parametrize("transport", ["rccl", "direct", "copy"])
def test_one_exchange_per_tvb_stage_gives_the_bits_of_the_two_of_the_reference(config, transport, monkeypatch):
    """src_mpi/limiter.cc:232 + src_mpi/claw.cc:793 merged (the default where ghost cells are known by their traces): the cut cells
    travel unlimited with their neighbours' averages, the receiver limits its ghost cells with the owner's inputs and arithmetic --
    against DFLO_TVB_ONE_EXCHANGE=0 (averages, rim limiter, limited traces): every bit"""
    if config == "c4":
        mesh, prm, ic, programs = L.case("c4")
    else:
        mesh, prm, ic = S._case("c3")
    out = []
    for one in ("1", "0"):
        monkeypatch.setenv("DFLO_TVB_ONE_EXCHANGE", one)
        claw = _self(mesh, prm, transport)
        what = claw.comm_info()[2]
        assert ("one exchange per stage" in what) == (one == "1") and ("two exchanges per stage" in what) == (one == "0"), what
        if config == "c4":
            L.setup(claw, mesh, ic, programs)
            out.append(L.run(claw, True))
        else:
            S._setup(claw, mesh, ic)
            out.append(S._run(claw, True))
        claw.close()
    a, b = out
    assert a["dt"] == b["dt"] and a["t"] == b["t"]
    assert np.array_equal(a["u"], b["u"]) and np.array_equal(a["avg"], b["avg"])


@pytest.mark.parametrize("config", ["c2", "c4", "c3"])
@pytest.mark.parametrize("transport", ["rccl", "direct"])
def test_the_compute_stream_ordered_by_a_wait_inside_the_interior_launch(config, transport, monkeypatch):
    """round 6: no wait packet between two kernels of the compute stream -- the interior launch's first workgroup waits at its end for a
    word the comm stream's pack kernel publishes (dflo_hip_stage_tail_wait / dflo_hip_pack_publish) -- against DFLO_TAIL_WAIT=0 (the
    event wait of round 5): the same bits, and the description says which"""
    if config == "c3":
        mesh, prm, ic = S._case("c3")
        mesh = dflo_amd.Mesh.cartesian(256, 64, 0.0, 0.0, 1.0 / 256, [2, 1, 0, 0], 1)     # (large enough for interior shards)
    else:
        mesh, prm, ic, programs = L.case(config)
    out = []
    for on in ("1", "0"):
        monkeypatch.setenv("DFLO_TAIL_WAIT", on)
        claw = _self(mesh, prm, transport)
        assert ("no wait packet" in claw.comm_info()[2]) == (on == "1"), claw.comm_info()[2]
        if config == "c3":
            S._setup(claw, mesh, ic)
            out.append(S._run(claw, True))
        else:
            L.setup(claw, mesh, ic, programs)
            out.append(L.run(claw, config == "c4"))
        claw.close()
    a, b = out
    assert a["dt"] == b["dt"] and a["t"] == b["t"]
    assert np.array_equal(a["u"], b["u"]) and np.array_equal(a["avg"], b["avg"])


SMALL = [("c2", "slab", "rccl"), ("c1", "slab", "rccl"), ("c1", "slab", "direct"), ("c3", "slab", "rccl"), ("c3", "slab", "direct"),
         ("c4", "slab", "rccl"), ("c5", "rcb", "rccl"), ("c5", "rcb", "direct"), ("kxrcf", "slab", "rccl"), ("kxrcf", "rcb", "direct"),
         ("pk", "slab", "rccl"), ("pkq1", "rcb", "direct"),
         ("c1", "slab", "ipc"), ("c3", "slab", "ipc"), ("c5", "rcb", "ipc"), ("kxrcf", "slab", "ipc"), ("pk", "slab", "ipc")]


@pytest.mark.parametrize("name,method,transport", SMALL)
def test_small_configurations_self_halo(name, method, transport):
    """every kind of record that travels: traces, traces + averages (LxF, TVB), whole cells (Pk, KXRCF)"""
    mesh, prm, ic = S._case(name)
    limited = prm.limiter == "TVB"
    one = dflo_amd.ConservationLaw(mesh, prm)
    S._setup(one, mesh, ic)
    ref = S._run(one, limited)
    one.close()
    claw = _self(mesh, prm, transport, method)
    S._setup(claw, mesh, ic)
    got = S._run(claw, limited)
    claw.close()
    if mesh.basis == "Pk":     # modal basis: where the shard rims lie shows in the last digits (see test_gpu_multi.py)
        assert np.allclose(got["dt"], ref["dt"], rtol=1e-12, atol=0)
        assert S.rel(got["u"], ref["u"]) < 1e-9 and S.rel(got["avg"], ref["avg"]) < 1e-10
        return
    assert got["dt"] == ref["dt"] and got["t"] == ref["t"]
    if limited or prm.pos_lim:
        assert S.rel(got["u"], ref["u"]) < 1e-9
    else:
        assert np.array_equal(got["u"], ref["u"]) and np.array_equal(got["avg"], ref["avg"])


def test_self_halo_refuses_a_mesh_without_a_cut():
    mesh = dflo_amd.Mesh.cartesian(16, 16, 0.0, 0.0, 1.0 / 16, [0, 0, 0, 0], 1)
    with pytest.raises(dflo_amd.DfloError, match="no periodic faces"):
        dflo_amd.MultiConservationLaw.for_self(mesh, dflo_amd.Parameters(flux="hllc", boundary={0: "outflow"}), 0, n_virtual=1)
