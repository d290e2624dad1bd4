"""The device arena's block list (csrc/arena_list.hpp: first fit, coalescing holes, blocks cut back to a part of themselves) on random sequences -- no GPU needed: the bookkeeping the
arena (csrc/arena.cpp) cuts every device block of the library with.  pg_host_emu_arena_blocks checks after every step that blocks are aligned,
inside the range and disjoint, that holes never touch (they merge), that holes + blocks = the range, and that the list is one hole again at the end."""
import ctypes as C

import pytest


@pytest.mark.parametrize("seed", [1, 2, 3, 20260927])
@pytest.mark.parametrize("size,max_block,n_ops", [(1 << 20, 1 << 14, 20000), (1 << 34, 1 << 31, 20000), (288 << 30, 100 << 30, 5000), (4096, 4096, 2000)])
def test_block_list_invariants_hold_on_random_sequences(seed, size, max_block, n_ops):
    from soapdenovo2_amd import api
    out = (C.c_uint64 * 4)()
    rc = api.lib().pg_host_emu_arena_blocks(seed, n_ops, size, max_block, out)
    assert rc == 0, f"invariant broken at step {-rc}"
    granted, refused, peak, holes = [int(v) for v in out]
    assert granted > 0 and peak <= size and holes >= 1
    if size >= (1 << 30):
        assert refused < granted                                          # (a nearly empty range refuses little)


def test_block_list_refuses_what_does_not_fit_and_reuses_what_came_back():
    from soapdenovo2_amd import api
    out = (C.c_uint64 * 4)()
    # a range of one page: blocks of up to the whole range -- most cuts are refused, the books still balance and the end is one hole
    assert api.lib().pg_host_emu_arena_blocks(7, 5000, 4096, 4096, out) == 0
    assert out[1] > 0 and out[2] <= 4096
