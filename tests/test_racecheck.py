"""The stream-ordering checker's core (racecheck.RaceCore) on synthetic launch logs, and its read / write table against the ABI."""
import os

from conftest import ROOT, load_sub


def _core():
    return load_sub("_lib").dev_tool("racecheck").RaceCore()


def _launch(c, s, reads=(), writes=(), what="k"):
    v = c.tick(s)
    for p, n in reads:
        c.access(v, s, p, n, False, what)
    for p, n in writes:
        c.access(v, s, p, n, True, what)


def test_missing_edge_is_reported_and_a_wait_removes_it():
    c = _core()
    _launch(c, "a", writes=[(1000, 64)], what="producer")
    _launch(c, "b", reads=[(1000, 64)], what="consumer")            # no edge a -> b
    assert [r["kind"] for r in c.reports.values()] == ["read-after-write"]
    c = _core()
    _launch(c, "a", writes=[(1000, 64)])
    c.wait_stream("b", "a")
    _launch(c, "b", reads=[(1000, 64)])
    assert not c.reports


def test_event_orders_only_what_preceded_its_record():
    c = _core()
    _launch(c, "a", writes=[(0x1000, 16)])
    c.record_event("e", "a")
    _launch(c, "a", writes=[(0x2000, 16)])                          # after the record
    c.wait_event("b", "e")
    _launch(c, "b", reads=[(0x1000, 16)])
    assert not c.reports
    _launch(c, "b", reads=[(0x2000, 16)])
    assert len(c.reports) == 1


def test_write_after_read_and_partial_overlap():
    c = _core()
    _launch(c, "a", reads=[(100, 100)])
    _launch(c, "b", writes=[(150, 10)])                             # overlaps the range a read
    assert [r["kind"] for r in c.reports.values()] == ["write-after-read"]
    c = _core()
    _launch(c, "a", reads=[(100, 100)])
    _launch(c, "b", writes=[(200, 10)])                             # adjacent, not overlapping
    assert not c.reports


def test_transitive_order_through_a_third_stream():
    c = _core()
    _launch(c, "a", writes=[(0, 8)])
    c.wait_stream("b", "a")
    _launch(c, "b", writes=[(64, 8)])
    c.wait_stream("c", "b")
    _launch(c, "c", reads=[(0, 8)])                                 # a -> b -> c
    assert not c.reports


def test_host_sync_orders_everything_issued_afterwards():
    c = _core()
    _launch(c, "a", writes=[(0, 8)])
    c.host_sync("a")
    _launch(c, "b", reads=[(0, 8)])
    assert not c.reports


def test_large_ranges_and_slices_of_them():
    c = _core()
    big = 64 << 20
    _launch(c, "lane", writes=[(4096 + 256, 1024)], what="wgrad into a slice of the arena")
    _launch(c, "main", reads=[(4096, big)], writes=[(4096, big)], what="adam over the arena")
    assert len(c.reports) == 2                                      # read and write of the arena both conflict with the slice
    c = _core()
    _launch(c, "lane", writes=[(4096 + 256, 1024)])
    c.wait_stream("main", "lane")
    _launch(c, "main", reads=[(4096, big)], writes=[(4096, big)])
    _launch(c, "main", writes=[(4096, big)], what="zero_grad")
    c.wait_stream("lane", "main")
    _launch(c, "lane", writes=[(4096 + 256, 1024)])
    assert not c.reports


def test_allocator_reuse_needs_an_order_or_record_stream():
    c = _core()
    c.new_storage(1 << 20, 4096, "a")
    _launch(c, "a", writes=[(1 << 20, 4096)])
    c.wait_stream("b", "a")
    _launch(c, "b", reads=[(1 << 20, 4096)], what="reader on another stream")
    # the tensor dies; the allocator hands the block to stream a's next allocation at once
    c.new_storage(1 << 20, 4096, "a")
    assert [r["kind"] for r in c.reports.values()] == ["reuse"]
    # announced to the allocator: fine
    c = _core()
    c.new_storage(1 << 20, 4096, "a")
    _launch(c, "a", writes=[(1 << 20, 4096)])
    c.wait_stream("b", "a")
    _launch(c, "b", reads=[(1 << 20, 4096)])
    c.record_stream((1 << 20) + 128, "b")
    c.new_storage(1 << 20, 2048, "a")
    assert not c.reports
    # or ordered: a waited for b before the block came back
    c = _core()
    c.new_storage(1 << 20, 4096, "a")
    c.wait_stream("b", "a")
    _launch(c, "b", reads=[(1 << 20, 4096)])
    c.wait_stream("a", "b")
    c.new_storage((1 << 20) - 1024, 8192, "a")                      # a merged, larger block
    assert not c.reports
    # the shadow of the old storage is forgotten: its accesses do not conflict with the new owner's
    _launch(c, "a", writes=[((1 << 20) - 1024, 8192)])
    assert not c.reports


def test_engine_handover_merges_the_producers_clock():
    c = _core()
    _launch(c, "main", writes=[(512, 64)], what="dy of a node on main")
    c.merge_from_writer("fork", 512)                                # the engine's wait before the consumer node runs on fork
    _launch(c, "fork", reads=[(512, 64)])
    assert not c.reports


def test_read_write_table_covers_the_abi():
    rc, lib = load_sub("_lib").dev_tool("racecheck"), load_sub("_lib")
    table = rc.parse_header(os.path.join(ROOT, "include", "sscg.h"))
    assert set(table) == set(lib.SIGNATURES)
    for name, row in table.items():
        assert len(row) == len(lib.SIGNATURES[name][1]), name
    kinds = dict(table["sscg_conv2d_wgrad"])
    assert kinds["x"] == "r" and kinds["dy"] == "r" and kinds["dw"] == "w" and kinds["ws"] == "w" and kinds["stream"] == "stream"
    kinds = dict(table["sscg_adam_step"])
    assert kinds["param"] == "w" and kinds["grad"] == "r" and kinds["shadow"] == "w"
    assert dict(table["sscg_norm_apply"])["residual"] == "r"
