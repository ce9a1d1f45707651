"""The read-back exchange of a group through RCCL, called from C++ inside libchunky_hip (chunky_group_transport; SURVEY.md
section 8e "a single RCCL reduce of per-tile radiance over xGMI"; the reference has one device and one queue,
RendererInstance.java:74-101, and reads back with clEnqueueReadBuffer, OpenClPathTracingRenderer.java:162-178).

What ONE GPU allows: (i) the real RCCL on a one-rank communicator, driven through the very code n GPUs run — with
CHUNKY_GROUP_SELF_EXCHANGE member 0's own blocks are packed, sent (to itself), received and scattered like any member's, and
the reduce form runs its ncclReduce; (ii) failure injection — members sharing a device (RCCL refuses the communicator), an RCCL
file that does not load or lacks a symbol, and a stand-in RCCL (tests/rccl_stub/, bound through CHUNKY_RCCL_LIB) that breaks
AFTER the communicator exists: at ncclSend, at ncclGroupEnd, or asynchronously.  In every case the render survives on peer
copies, the image is bit for bit the reference's golden image, and chunky_group_transport says why."""
import json
import os
import subprocess
import sys

import pytest

from chunkyclplugin_amd import native

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def child(devices, scene="outdoor", then=None, **env):
    e = dict(os.environ)
    for k in ("CHUNKY_RCCL_LIB", "CHUNKY_GROUP_TRANSPORT", "CHUNKY_GROUP_SELF_EXCHANGE", "CHUNKY_RCCL_TRY_SHARED", "RCCL_STUB_MODE",
              "CHUNKY_GROUP_NO_PROBE", "CHUNKY_GROUP_TIMEOUT_MS", "CHUNKY_HIP_LIB"):
        e.pop(k, None)
    if not env.pop("shipping", False):
        # the rig variables exist in the -DCHUNKY_TUNING build only (native.build_tuning): the shipping library reads none of them
        e["CHUNKY_HIP_LIB"] = native.build_tuning()
    e.update({k: str(v) for k, v in env.items()})
    cmd = [sys.executable, os.path.join(HERE, "rccl_child.py"), devices, scene] + ([str(then)] if then is not None else [])
    proc = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=e)
    assert proc.returncode == 0, (proc.stdout[-2000:], proc.stderr[-3000:])
    return json.loads([ln for ln in proc.stdout.splitlines() if ln.startswith("{")][-1])  # (RCCL prints lines of its own)


@pytest.fixture(scope="module")
def stub(tmp_path_factory):
    d = tmp_path_factory.mktemp("rccl_stub")
    src = os.path.join(HERE, "rccl_stub", "rccl_stub.c")
    full, old = str(d / "librccl_stub.so"), str(d / "librccl_stub_old.so")
    subprocess.run(["gcc", "-shared", "-fPIC", "-O1", src, "-o", full], check=True)
    subprocess.run(["gcc", "-shared", "-fPIC", "-O1", "-DRCCL_STUB_OMIT_REDUCE", src, "-o", old], check=True)
    return {"full": full, "old": old}


@pytest.mark.parametrize("scene", ["outdoor", "entities"])
def test_real_rccl_one_rank_send_recv(scene):
    out = child("0", scene, CHUNKY_GROUP_SELF_EXCHANGE=1)
    assert out["before"]["name"] == "rccl-sendrecv" and out["before"]["backend"] == "rccl", out
    assert "1 rank(s)" in out["before"]["detail"] and "rccl 2." in out["before"]["detail"], out
    assert out["identical"] and out["identical_again"] and out["first_nonzero"]
    assert out["after"]["name"] == "rccl-sendrecv"   # no fallback happened


def test_real_rccl_one_rank_reduce_and_switching():
    out = child("0", "outdoor", then=native.TRANSPORT_RCCL_REDUCE, CHUNKY_GROUP_SELF_EXCHANGE=1)
    assert out["before"]["name"] == "rccl-sendrecv" and out["switched"]["name"] == "rccl-reduce", out
    assert "ncclReduce" in out["switched"]["detail"]
    assert out["identical"] and out["identical_again"] and out["after"]["name"] == "rccl-reduce"
    out = child("0", "outdoor", then=native.TRANSPORT_PEER_COPY, CHUNKY_GROUP_TRANSPORT="rccl-reduce")
    assert out["before"]["name"] == "rccl-reduce" and out["switched"]["name"] == "peer-copy", out
    assert out["identical"] and out["identical_again"]


def test_members_sharing_a_device_use_peer_copies():
    out = child("0,0,0")
    assert out["before"]["name"] == "peer-copy" and "share device 0" in out["before"]["detail"], out
    assert out["identical"] and out["identical_again"]
    # asking for RCCL there is a state error, and the render goes on
    out = child("0,0,0", then=native.TRANSPORT_RCCL_SENDRECV)
    assert out["switch_error"]["code"] == native.E_STATE and "no RCCL communicator" in out["switch_error"]["message"], out
    assert out["identical"] and out["identical_again"] and out["after"]["name"] == "peer-copy"


def test_real_rccl_refuses_duplicate_devices_and_the_run_survives():
    out = child("0,0", CHUNKY_RCCL_TRY_SHARED=1)
    assert out["before"]["name"] == "peer-copy" and "ncclCommInitAll" in out["before"]["detail"], out
    assert out["identical"] and out["identical_again"]


def test_rccl_that_does_not_load(stub):
    out = child("0,0,0", CHUNKY_RCCL_LIB="/nonexistent/librccl.so", CHUNKY_RCCL_TRY_SHARED=1)
    assert out["before"]["name"] == "peer-copy" and "RCCL not loaded" in out["before"]["detail"], out
    assert out["identical"] and out["identical_again"]
    out = child("0,0,0", CHUNKY_RCCL_LIB=stub["old"], CHUNKY_RCCL_TRY_SHARED=1)
    assert out["before"]["name"] == "peer-copy" and "RCCL lacks ncclReduce" in out["before"]["detail"], out
    assert out["identical"] and out["identical_again"]


@pytest.mark.parametrize("mode", ["fail_send", "fail_end", "async", "ok"])
def test_first_contact_probe_catches_a_broken_rccl_at_group_creation(stub, mode):
    """chunky_group_create sends a known pattern from every member to member 0 through the new communicators and compares the
    bytes: an RCCL that returns an error, reports one asynchronously, or — mode "ok" — claims success and moves nothing never
    becomes the group's transport.  The reason is in chunky_group_transport, the render runs on peer copies."""
    out = child("0,0,0", CHUNKY_RCCL_LIB=stub["full"], CHUNKY_RCCL_TRY_SHARED=1, RCCL_STUB_MODE=mode)
    assert out["before"]["name"] == "peer-copy" and "failed its first exchange" in out["before"]["detail"], out
    assert ("stub" in out["before"]["detail"]) if mode != "ok" else ("arrived as 0" in out["before"]["detail"]), out
    assert out["identical"] and out["identical_again"] and out["first_nonzero"]


def test_real_rccl_passes_the_probe_in_the_shipping_library():
    """The library as shipped (no rig variable is read): a one-member group owns a real communicator, its first exchange —
    a send to itself — is verified, and RCCL is the transport."""
    out = child("0", "outdoor", shipping=True)
    assert out["before"]["name"] == "rccl-sendrecv" and "first exchange verified" in out["before"]["detail"], out
    assert out["identical"] and out["identical_again"]
    # ... and the rig variables mean nothing to it: asking for the reduce through the environment changes nothing
    out = child("0", "outdoor", shipping=True, CHUNKY_GROUP_TRANSPORT="rccl-reduce")
    assert out["before"]["name"] == "rccl-sendrecv", out


@pytest.mark.parametrize("transport", ["rccl", "rccl-reduce"])
def test_an_exchange_that_never_finishes_is_aborted_not_waited_for(transport):
    """A deadline of 0 ms makes the REAL RCCL's exchange "hang": the library must not sit in hipStreamSynchronize behind the
    collective's kernel — it polls, gives up, calls ncclCommAbort FIRST (the one call that ends a stuck RCCL kernel), then drains
    the streams and repeats the read-back on peer copies.  The image is the reference's."""
    out = child("0", "outdoor", CHUNKY_GROUP_SELF_EXCHANGE=1, CHUNKY_GROUP_NO_PROBE=1, CHUNKY_GROUP_TIMEOUT_MS=0, CHUNKY_GROUP_TRANSPORT=transport)
    assert out["before"]["backend"] == "rccl", out
    assert out["after_first"]["name"] == "peer-copy" and "unfinished after 0 ms" in out["after_first"]["detail"], out
    assert out["identical"] and out["identical_again"] and out["first_nonzero"]


@pytest.mark.parametrize("mode,transport", [("fail_send", "rccl"), ("fail_end", "rccl"), ("async", "rccl"),
                                            ("fail_send", "rccl-reduce"), ("fail_end", "rccl-reduce"), ("async", "rccl-reduce")])
def test_rccl_that_breaks_after_the_communicator_exists(stub, mode, transport):
    """The communicator was created (the stand-in accepts anything; the first-contact probe is switched off here), then the first
    exchange fails: that read-back and every later one run on peer copies, nothing of the render is lost."""
    out = child("0,0,0", CHUNKY_RCCL_LIB=stub["full"], CHUNKY_RCCL_TRY_SHARED=1, RCCL_STUB_MODE=mode, CHUNKY_GROUP_TRANSPORT=transport, CHUNKY_GROUP_NO_PROBE=1)
    assert out["before"]["backend"] == "rccl" and "3 rank(s)" in out["before"]["detail"], out
    assert out["after_first"]["name"] == "peer-copy" and "stub" in out["after_first"]["detail"], out
    assert out["identical"] and out["identical_again"] and out["first_nonzero"]
