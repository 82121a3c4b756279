"""BASELINE.json configurations at their full sizes on the MI355X, through size-independent properties
(round-trip bounds, batch-split invariance, sampled frames against the oracle).  GPU only."""
import numpy as np
import pytest

from frad_python_amd import synth
from oracle import frad_oracle as fo

pytestmark = pytest.mark.gpu
EPS64, EPS32 = 2.220446049250313e-16, 1.1920929e-07


@pytest.fixture(scope="module")
def gpu():
    import torch
    from frad_python_amd import core
    assert torch.cuda.is_available()
    return torch, core, torch.device("cuda:0")


def _signal(torch, dev, n, C, seed, dtype):
    g = torch.Generator(device=dev).manual_seed(seed)
    t = torch.arange(n, dtype=torch.float64, device=dev) / 48000
    x = torch.stack([sum(torch.sin(2 * np.pi * 110.0 * (c + 1) * (h + 1) * t + h) / (h + 1) for h in range(4)) for c in range(C)], 1)
    x = x * (0.8 / x.abs().max()) + torch.randn((n, C), generator=g, device=dev, dtype=torch.float64) * 1e-3
    if dtype == "s16le":
        return torch.clamp(torch.round(x * 32768), -32768, 32767).to(torch.int16)
    return (x * 0.9).to(torch.float32)


def test_cfg2_ten_minutes_stereo_profile0(gpu):
    torch, core, dev = gpu
    N, C, F = 2048, 2, 14062
    pcm = _signal(torch, dev, F * N, C, 1234, "s16le")
    for bits, tol in ((32, 2.0 ** -23), (64, 0.0), (16, 2.0 ** -10)):
        enc = core.analogue_batch(0, pcm, "s16le", F, N, C, bits)
        assert not enc.escalated and bool(torch.isfinite(enc.absmax).all())
        dec = core.digital_batch(0, enc.payload, F, N, C, bits)
        x = pcm.to(torch.float64).reshape(F, N, C) / 32768
        # round trip: storage rounding of every coefficient (relative `tol`) + transform rounding
        bound = tol * float(enc.absmax.max()) * np.sqrt(N) * 4 + 64 * EPS64 * 11
        assert float((dec - x).abs().max()) <= bound, bits
        # batch-split invariance = sharding invariance: two halves give the same bits as one launch
        h = F // 2
        a = core.analogue_batch(0, pcm[:h * N], "s16le", h, N, C, bits).payload
        b = core.analogue_batch(0, pcm[h * N:], "s16le", F - h, N, C, bits).payload
        assert torch.equal(torch.cat([a, b]), enc.payload)
        # sampled frames against the oracle
        pay = enc.payload[:, :enc.nbytes]
        host = pcm.cpu().numpy()
        mism = words = 0
        for f in (0, 1, 777, 7031, 14060, 14061):
            frame = fo.to_f64(host[f * N:(f + 1) * N], fo.pcm_dtype("s16le"))
            want = fo.pack_floats(fo.dct_channels(frame).T.ravel(), bits, False)
            got = pay[f].cpu().numpy()
            if bits <= 32:                                     # stored words (not bytes): a rounding tie changes one word
                wd = np.dtype(">u%d" % (bits // 8))
                mism += int(np.count_nonzero(np.frombuffer(want, wd) != np.frombuffer(got.tobytes(), wd)))
                words += N * C
            else:
                gv, wv = fo.unpack_floats(got.tobytes(), bits, False), fo.unpack_floats(want, bits, False)
                assert np.max(np.abs(gv - wv)) <= 8 * EPS64 * np.max(np.abs(wv)) * 11
            ref = fo.p0_digital(want, fo.DEPTHS.index(bits), C, False)
            one = core.digital_batch(0, torch.from_numpy(np.frombuffer(want, np.uint8).copy()).to(dev).reshape(1, -1), 1, N, C, bits)
            assert np.max(np.abs(one[0].cpu().numpy() - ref)) <= 8 * EPS64 * 11
        if bits <= 32:                                         # the word contract of DESIGN.md section 5: <= max(2, 1e-5 x words)
            print(f"[cfg2 full size] bits={bits}: {mism} of {words} sampled stored words differ from the oracle's")
            assert mism <= max(2, 1e-5 * words)


def test_cfg3_clip_batch_sharded_like_eight_gpus(gpu):
    torch, core, dev = gpu
    from frad_python_amd.parallel import shard_range
    clips, n, C, N = 512, 48000, 2, 2048                      # one GPU's share of the 4096-clip batch
    full, tail = n // N, n % N                                # 23 frames + 896-sample tail per clip
    pcm = _signal(torch, dev, clips * n, C, 7, "s16le").reshape(clips, n, C)
    body = pcm[:, :full * N].contiguous()
    enc = core.analogue_batch(0, body, "s16le", clips * full, N, C, 32)
    parts = []
    for r in range(8):                                        # contiguous clip ranges, no exchange between them
        a, b = shard_range(clips, r, 8)
        parts.append(core.analogue_batch(0, body[a:b].contiguous(), "s16le", (b - a) * full, N, C, 32).payload)
    assert torch.equal(torch.cat(parts), enc.payload)
    # tails: 896 = 7 * 128 samples -> the any-N kernel, one launch over all clips
    tails = pcm[:, full * N:].contiguous()
    et = core.analogue_batch(0, tails, "s16le", clips, tail, C, 32)
    dt = core.digital_batch(0, et.payload, clips, tail, C, 32)
    assert float((dt - tails.to(torch.float64) / 32768).abs().max()) <= 1e-5
    frame = fo.to_f64(tails[3].cpu().numpy(), fo.pcm_dtype("s16le"))
    want = fo.unpack_floats(fo.pack_floats(fo.dct_channels(frame).T.ravel(), 32, False), 32, False)
    got = fo.unpack_floats(et.payload[3, :et.nbytes].cpu().numpy().tobytes(), 32, False)
    assert np.max(np.abs(got - want)) <= 2.0 ** -22 * np.max(np.abs(want))
    # the same batch consumed and produced IN PLACE (frad_p0_analogue_clips / _digital_clips): [clips, 48000, C] as it is
    # resident, no gathered copies -- bit-identical to the gathered batches above
    ec = core.analogue_clips(pcm, "s16le", N, 32)
    assert torch.equal(ec.payload, enc.payload) and torch.equal(ec.absmax, enc.absmax)
    etc = core.analogue_clips(pcm, "s16le", tail, 32, first=full * N)
    assert torch.equal(etc.payload, et.payload) and torch.equal(etc.absmax, et.absmax)
    whole = torch.full((clips, n, C), float("nan"), dtype=torch.float64, device=dev)
    core.digital_clips(ec.payload, whole, N, 32)
    core.digital_clips(etc.payload, whole, tail, 32, first=full * N)
    db = core.digital_batch(0, enc.payload, clips * full, N, C, 32)
    assert torch.equal(whole[:, :full * N].reshape(clips * full, N, C), db)
    assert torch.equal(whole[:, full * N:], dt)


def test_clip_batches_of_other_geometries(gpu):
    """frad_p0_*_clips beyond the kernels that address clips themselves: every geometry gives the flat batch's bits
    (N = 1024 unit kernels, N = 4096 eight-channel float32, an odd frame length with float32 PCM, mono N = 2048 with an
    odd number of frames per clip)."""
    torch, core, dev = gpu
    for (N, C, fmt, clip_len, n_clips) in ((1024, 2, "s16le", 5000, 9), (4096, 8, "f32le", 9000, 3), (300, 3, "f32le", 1000, 5),
                                           (2048, 1, "s16le", 7 * 2048 + 8, 5), (2048, 2, "s32le", 3 * 2048 + 4, 4)):
        pcm = _signal(torch, dev, n_clips * clip_len, C, N + C, fmt).reshape(n_clips, clip_len, C)
        fpc = clip_len // N
        body = pcm[:, :fpc * N].contiguous()
        flat = core.analogue_batch(0, body, fmt, n_clips * fpc, N, C, 32, check_overflow=False)
        ec = core.analogue_clips(pcm, fmt, N, 32)
        assert torch.equal(ec.payload, flat.payload) and torch.equal(ec.absmax, flat.absmax), (N, C, fmt)
        out = torch.zeros((n_clips, clip_len, C), dtype=torch.float64, device=dev)
        core.digital_clips(ec.payload, out, N, 32)
        want = core.digital_batch(0, flat.payload, n_clips * fpc, N, C, 32)
        assert torch.equal(out[:, :fpc * N].reshape(n_clips * fpc, N, C), want), (N, C, fmt)
        assert float(out[:, fpc * N:].abs().max()) == 0.0       # nothing written behind the last whole frame


def test_cfg4_192k_eight_channel_float32(gpu):
    torch, core, dev = gpu
    N, C, F = 4096, 8, 2812
    pcm = _signal(torch, dev, F * N, C, 99, "f32le")
    enc = core.analogue_batch(0, pcm, "f32le", F, N, C, 32)        # f32 compute, as the reference does for float PCM
    dec = core.digital_batch(0, enc.payload, F, N, C, 32)          # decode is float64 (channel-group kernel)
    assert float((dec - pcm.to(torch.float64).reshape(F, N, C)).abs().max()) <= 64 * EPS32
    host = pcm.cpu().numpy()
    for f in (0, 1406, 2811):
        X = fo.dct_channels(host[f * N:(f + 1) * N])
        assert X.dtype == np.float32
        want = X.T.ravel().astype(np.float64)
        got = fo.unpack_floats(enc.payload[f, :enc.nbytes].cpu().numpy().tobytes(), 32, False)
        assert np.max(np.abs(got - want)) <= 8 * EPS32 * np.max(np.abs(want)) * 12
        assert abs(float(enc.absmax[f]) - float(np.max(np.abs(X)))) <= 8 * EPS32 * float(np.max(np.abs(X))) * 12
        ref = fo.p0_digital(enc.payload[f, :enc.nbytes].cpu().numpy().tobytes(), 3, C, False)
        assert np.max(np.abs(dec[f].cpu().numpy() - ref)) <= 8 * EPS64 * 12


def test_cfg5_lossy_profile_sixty_seconds(gpu):
    torch, core, dev = gpu
    N, C, hop, srate = 2048, 2, 1920, 48000
    n = 60 * srate
    F = (n - N) // hop + 1
    pcm = _signal(torch, dev, n, C, 1234, "s16le")
    loss = 1.25 ** 20 / 19.0 + 0.5                                   # --losslevel 20
    q, tq = core.p1_analogue_batch(pcm, "s16le", F, N, C, 16, srate, loss, frame_stride=hop)
    dec = core.p1_digital_batch(q, tq, N, C, 16, srate)
    out, tail = core.p1_overlap_add(dec, 16)
    host = pcm.cpu().numpy()
    differing, total, err = 0, 0, 0.0
    for f in (0, 1, 750, F - 1):
        wq, wt, aux = fo.p1_analogue_pre(fo.to_f64(host[f * hop:f * hop + N], fo.pcm_dtype("s16le")), 16, srate, loss)
        gq, gt = q[f].cpu().numpy().reshape(-1), tq[f].cpu().numpy().reshape(-1)
        assert np.abs(gq - wq).max() <= 1 and np.abs(gt - wt).max() <= 1
        differing += int(np.count_nonzero(gq != wq)); total += gq.size
        ref = fo.p1_digital_post(gq, gt, 2, C, srate, N)
        err = max(err, float(np.max(np.abs(dec[f].cpu().numpy() - ref))))
    assert differing <= 1e-3 * total and err <= 1e-12
    # PSNR of the build-decoded PCM against the oracle-decoded PCM (same integers through the reference math,
    # incl. the Hann cross-fade) over the first 20 frames, and the lossy quality itself for the record
    ola, ref_out = fo.OverlapAdd(), []
    for f in range(20):
        wq, wt, aux = fo.p1_analogue_pre(fo.to_f64(host[f * hop:f * hop + N], fo.pcm_dtype("s16le")), 16, srate, loss)
        ref_out.append(ola.push(fo.p1_digital_post(wq, wt, 2, C, srate, N), True, 16))
    ref_out = np.concatenate(ref_out)
    got = out[:20].reshape(-1, C).cpu().numpy()
    mse = float(np.mean((got - ref_out) ** 2))
    assert mse == 0.0 or 10 * np.log10(1.0 / mse) > 100, "build-decoded vs reference-decoded PSNR"
    x = host[:20 * hop].astype(np.float64) / 32768
    lossy_psnr_build = 10 * np.log10(1.0 / np.mean((x[hop:] - got[hop:]) ** 2))
    lossy_psnr_ref = 10 * np.log10(1.0 / np.mean((x[hop:] - ref_out[hop:]) ** 2))
    assert abs(lossy_psnr_build - lossy_psnr_ref) < 0.01           # the codec's own loss (~19 dB at level 20) is unchanged


def test_step_is_hip_graph_capturable(gpu):
    """After frad_plan_prepare (and one eager call that settles the kernels' LDS attributes) the launches
    make no allocation or synchronising call, so a whole encode -> overflow scan -> decode step can be
    captured in a HIP graph and replayed (include/frad_hip.h: frad_plan_prepare)."""
    torch, core, dev = gpu
    N, C, F, bits = 2048, 2, 512, 32
    core._lib.load().plan_prepare(N, False)
    pcm = _signal(torch, dev, F * N, C, 7, "s16le")
    pay = torch.empty((F, core._lib.load().payload_bytes(N, C, bits)), dtype=torch.uint8, device=dev)
    am = torch.empty(F, dtype=torch.float64, device=dev)
    out = torch.empty((F, N, C), dtype=torch.float64, device=dev)
    flag = torch.zeros((), dtype=torch.int32, device=dev)

    def step():
        core.analogue_batch(0, pcm, "s16le", F, N, C, bits, check_overflow=False, out=pay, absmax=am)
        core.overflow_scan(am, bits, flag)
        core.digital_batch(0, pay, F, N, C, bits, out=out)
    step()
    torch.cuda.synchronize()
    want_pay, want_out = pay.clone(), out.clone()
    side = torch.cuda.Stream(device=dev)
    side.wait_stream(torch.cuda.current_stream())
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        with torch.cuda.graph(graph, stream=side):
            step()
    torch.cuda.synchronize()
    for _ in range(2):
        pay.zero_(); out.zero_()
        graph.replay()
        torch.cuda.synchronize()
        assert torch.equal(pay, want_pay) and torch.equal(out, want_out)
    assert int(flag.item()) == 0
