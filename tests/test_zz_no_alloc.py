"""The entry point that allocates nothing (rg_pack_reads_noalloc, include/raygraph.h): its output against the
existing read ingest (rg_add_reads_text + rg_get_reads) through the C ABI, bit for bit, on two calls in a row,
the second with a smaller input (the library's scratch array keeps the first call's contents)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _raw_sequences(n, seed, maxlen=160):
    rng = np.random.default_rng(seed)
    alphabet = np.frombuffer(b"ACGTacgtNnxR-", dtype=np.uint8)
    weights = np.array([20, 20, 20, 20, 3, 3, 3, 3, 2, 1, 1, 1, 1], dtype=np.float64)
    weights /= weights.sum()
    lens = rng.integers(0, maxlen, size=n).astype(np.uint32)
    lens[::17] = 0                                  # empty sequences
    text = rng.choice(alphabet, size=int(lens.sum()) + 8, p=weights).astype(np.uint8)
    starts = np.zeros(n, dtype=np.uint64)
    starts[1:] = np.cumsum(lens[:-1], dtype=np.uint64)
    for i in range(0, n, 11):                       # sequences that trim to nothing
        text[int(starts[i]):int(starts[i]) + int(lens[i])] = ord("N")
    return text, starts, lens


def _existing_entry_point(text, starts, lens):
    import ray_amd
    b = ray_amd.GraphBuilder(21)
    b.add_reads_text(text.tobytes(), starts, lens)
    out = b.get_reads()
    b.close()
    return out


def _no_alloc(b, text, starts, lens, cap_reads, out):
    import torch
    n = len(lens)
    d_text = torch.from_numpy(text).cuda()
    d_starts = torch.from_numpy(starts.astype(np.int64)).cuda()
    d_lens = torch.from_numpy(lens.astype(np.int32)).cuda()
    torch.cuda.synchronize()
    d_packed, d_off, d_len = out
    free_before = torch.cuda.mem_get_info()[0]        # hipMemGetInfo: what the driver has left on the device
    b.pack_reads_noalloc(d_text.data_ptr(), len(text), d_starts.data_ptr(), d_lens.data_ptr(), n, d_packed.data_ptr(),
                         d_packed.numel(), d_off.data_ptr(), d_len.data_ptr())
    torch.cuda.synchronize()                          # the builder runs on its own stream
    assert torch.cuda.mem_get_info()[0] == free_before, "rg_pack_reads_noalloc changed the device's free memory"
    h_off = d_off.cpu().numpy().astype(np.uint64)
    total = int(h_off[n])
    return d_packed.cpu().numpy()[:total], h_off[:n], d_len.cpu().numpy().astype(np.uint32)[:n], total


def test_no_alloc_ingest_matches_add_reads_text_twice():
    import torch
    import ray_amd
    from ray_amd._lib import RG_NOALLOC_MAX_READS
    big = _raw_sequences(50000, 5)
    small = _raw_sequences(3000, 6, maxlen=90)
    cap_reads = 50000
    # the caller's buffers, allocated once, filled with a pattern so that unwritten bytes show
    d_packed = torch.full((cap_reads * 40 + 64,), 0xEE, dtype=torch.uint8, device="cuda")
    d_off = torch.full((cap_reads + 1,), -1, dtype=torch.int64, device="cuda")
    d_len = torch.full((cap_reads,), -1, dtype=torch.int32, device="cuda")
    with ray_amd.GraphBuilder(21) as b:
        for text, starts, lens in (big, small):
            want_p, want_o, want_l = _existing_entry_point(text, starts, lens)
            d_packed.fill_(0xEE)
            got_p, got_o, got_l, total = _no_alloc(b, text, starts, lens, cap_reads, (d_packed, d_off, d_len))
            assert total == len(want_p)
            assert np.array_equal(got_l, want_l), "stored lengths differ"
            assert np.array_equal(got_o, want_o), "byte offsets differ"
            assert np.array_equal(got_p, want_p), "packed bases differ"
            assert (want_l > 0).sum() > len(lens) // 2 and (want_l == 0).sum() > 0
        # a buffer that is too small: nothing beyond it is written, the total still tells what was needed
        text, starts, lens = small
        want_p, _, _ = _existing_entry_point(text, starts, lens)
        d_packed.fill_(0xEE)
        cut = len(want_p) // 2
        t_text, t_starts, t_lens = torch.from_numpy(text).cuda(), torch.from_numpy(starts.astype(np.int64)).cuda(), torch.from_numpy(lens.astype(np.int32)).cuda()
        torch.cuda.synchronize()
        b.pack_reads_noalloc(t_text.data_ptr(), len(text), t_starts.data_ptr(), t_lens.data_ptr(), len(lens), d_packed.data_ptr(), cut,
                             d_off.data_ptr(), d_len.data_ptr())
        torch.cuda.synchronize()
        h = d_packed.cpu().numpy()
        assert int(d_off[len(lens)].item()) == len(want_p)
        assert np.array_equal(h[:cut], want_p[:cut]) and (h[cut:cut + 64] == 0xEE).all()
        # a start or a length that leaves the text (the arrays are device memory, so the check is the kernel's): read as an
        # empty sequence, nothing outside the text is touched; the others are unaffected
        bad_starts, bad_lens = starts.copy(), lens.copy()
        victims = [i for i in range(5, len(lens), 97) if lens[i] > 0][:6]
        bad_starts[victims[0]] = len(text) + 12345            # start beyond the text
        bad_starts[victims[1]] = (1 << 62)                    # absurd start
        bad_lens[victims[2]] = 70000                          # longer than a Read can store
        bad_starts[victims[3]] = len(text) - 3                # starts inside, ends outside
        bad_lens[victims[3]] = 50
        fixed_lens = lens.copy()
        fixed_lens[victims[:4]] = 0
        want_p, want_o, want_l = _existing_entry_point(text, starts, fixed_lens)
        d_packed.fill_(0xEE)
        got_p, got_o, got_l, total = _no_alloc(b, text, bad_starts, bad_lens, cap_reads, (d_packed, d_off, d_len))
        assert total == len(want_p) and np.array_equal(got_l, want_l) and np.array_equal(got_o, want_o) and np.array_equal(got_p, want_p)
        assert (got_l[victims[:4]] == 0).all()
        # beyond the scratch's stated maximum: a clear error, nothing queued
        with pytest.raises(ray_amd.RayGraphError) as e:
            b.pack_reads_noalloc(0, 0, 0, 0, RG_NOALLOC_MAX_READS + 1, 0, 0, d_off.data_ptr(), 0)
        assert e.value.code == -5 and "scratch holds" in str(e.value)
        # no sequences at all: zero packed bytes
        b.pack_reads_noalloc(0, 0, 0, 0, 0, 0, 0, d_off.data_ptr(), 0)
        torch.cuda.synchronize()
        assert int(d_off[0].item()) == 0
