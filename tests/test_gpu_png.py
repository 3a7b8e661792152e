"""GPU suite: the PNG encoder kernels (A9 on the device, csrc/kernels_png.hip) through the C ABI -- byte for byte against the
lane-level model (oracle/png_model.py), decoded by PIL back to the exact 8-bit frame, from u8 and from float sources, and as the
stream-level call fav_stylize uses."""
import io
import os

import numpy as np
import pytest

from fav_amd import synth

pytestmark = pytest.mark.gpu


def T(a, dev):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def _decode(data):
    from PIL import Image
    return np.array(Image.open(io.BytesIO(data)).convert("RGB"))


@pytest.mark.parametrize("shape,kind", [((40, 97), "smooth"), ((17, 70), "noise"), ((9, 130), "flat"), ((1, 1), "noise"), ((5, 300), "gradient"),
                                        ((6, 333), "runs"), ((4, 33), "high"), ((3, 100), "short-runs"), ((2, 4001), "noise"), ((64, 64), "smooth")])
def test_png_bytes_equal_the_model(favlib, oracle, cuda, shape, kind):
    import png_model as P
    h, w = shape
    rng = np.random.default_rng(h * 1000 + w)
    if kind == "smooth": img = synth.smooth_frame(h, w, 3)
    elif kind == "flat": img = np.full((h, w, 3), 200, np.uint8)
    elif kind == "gradient": img = np.tile((np.arange(w) % 256).astype(np.uint8)[None, :, None], (h, 1, 3))
    elif kind == "runs": img = np.repeat(rng.integers(0, 256, (h, (w + 36) // 37, 3), dtype=np.uint8), 37, axis=1)[:, :w]
    elif kind == "high": img = rng.integers(144, 256, (h, w, 3), dtype=np.uint8)
    elif kind == "short-runs": img = np.repeat(rng.integers(0, 256, (h, w // 2, 3), dtype=np.uint8), 2, axis=1)
    else: img = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    img = np.ascontiguousarray(img)
    got = favlib.png_encode(T(img, cuda))
    want = P.encode(img)
    assert got == want, f"{len(got)} vs {len(want)} bytes; first difference at {next((i for i, (a, b) in enumerate(zip(got, want)) if a != b), -1)}"
    assert np.array_equal(_decode(got), img)


def test_png_from_float_state_and_stream(favlib, oracle, cuda, golden_dir):
    """quantisation fused into the encoder (clamp, x255, truncate = quantize_kernel / image.save) and the stream-level entry point"""
    import torch
    rng = np.random.default_rng(9)
    h, w = 37, 91
    f = (rng.standard_normal((3, h, w)) * 0.5 + 0.5).astype(np.float32)
    f[0, 0, :4] = [0.0, 1.0, 254.999 / 255, -3.0]
    want8 = oracle.to_u8_hwc(f)
    got = favlib.png_encode(T(f, cuda))
    assert np.array_equal(_decode(got), want8)
    # the stream's current frame: identical pixels to the u8 output of the same call
    net = favlib.Net(os.path.join(golden_dir, "tiny_model.t7"), 0)
    hh, ww = 48, 64
    st = favlib.Stream(net, hh, ww)
    _, u8 = st.first_frame(T(synth.smooth_frame(hh, ww, 1), cuda), want_u8=True)
    png = favlib.png_encode(None, from_stream=st)
    assert np.array_equal(_decode(png), u8.cpu().numpy())
    assert png == favlib.png_encode(u8.contiguous())          # the u8 and the float source give the same file


@pytest.mark.parametrize("size", [(48, 64), (180, 320)])
def test_png_encoded_next_to_the_following_frames(favlib, oracle, cuda, golden_dir, size):
    """fav_stream_encode_png_async: every frame's file produced on the stream's encoder queue while the NEXT frames are already being
    stylised (double-buffered state: frame i + 1 is written next to the one being encoded, frame i + 2 waits for that buffer's encoder)
    -- same bytes as the encode on the compute queue of a second stream fed the same inputs, same recurrent states, and
    set_state / a synchronous encode in between stay ordered."""
    import torch
    net = favlib.Net(os.path.join(golden_dir, "tiny_model.t7"), 0)
    H, W = size
    N = 9
    fr = [T(synth.random_frame(H, W, 40 + i), cuda) for i in range(N)]
    bw = [T(synth.backward_flow(H, W, 50 + i), cuda) for i in range(N)]
    fw = [T(synth.forward_flow_from_backward(bw[i].cpu().numpy(), 60 + i), cuda) for i in range(N)]
    a, b = favlib.Stream(net, H, W), favlib.Stream(net, H, W)
    want, states = [], []
    b.first_frame(fr[0], want_f32=False)
    for i in range(N):
        if i: b.next_frame_flow(fr[i], bw[i], fw[i], use_structure=False, want_f32=False)
        want.append(favlib.png_encode(None, from_stream=b)); states.append(b.state().cpu().numpy())
    bufs = [a.png_buffers() for _ in range(N)]
    a.first_frame(fr[0], want_f32=False)
    for i in range(N):                                   # nothing waits between the frames
        if i: a.next_frame_flow(fr[i], bw[i], fw[i], use_structure=False, want_f32=False)
        a.encode_png_async_into(*bufs[i])
    a.wait_png()
    torch.cuda.synchronize()
    for i in range(N):
        n = int(bufs[i][1].item())
        assert bufs[i][0][:n].cpu().numpy().tobytes() == want[i], "frame %d" % i
    assert np.array_equal(a.state().cpu().numpy(), states[-1])
    # a synchronous encode and set_state after asynchronous ones
    assert favlib.png_encode(None, from_stream=a) == want[-1]
    a.encode_png_async_into(*bufs[0])
    a.set_state(T(states[3], cuda))                      # (into the buffer the encoder is reading: waits for it on the device)
    a.next_frame_flow(fr[4], bw[4], fw[4], use_structure=False, want_f32=False)
    a.encode_png_async_into(*bufs[1])
    a.wait_png(); torch.cuda.synchronize()
    assert bufs[0][0][:int(bufs[0][1].item())].cpu().numpy().tobytes() == want[-1]
    assert bufs[1][0][:int(bufs[1][1].item())].cpu().numpy().tobytes() == want[4]


def test_png_full_size_1280x720_decodes_exactly(favlib, oracle, cuda):
    """BASELINE config 3 size: decodes (PIL) to the exact bytes, every repeated call gives the same file, size within the capacity"""
    import png_model as P
    import torch
    h, w = 720, 1280
    rng = np.random.default_rng(1)
    img = synth.smooth_frame(h, w, 8)
    img[100:200, 300:900] = 17                       # a flat patch: long runs
    img[300:400] = rng.integers(0, 256, (100, w, 3), dtype=np.uint8)
    a = favlib.png_encode(T(img, cuda))
    b = favlib.png_encode(T(img, cuda))
    assert a == b and len(a) <= P.capacity(w, h)
    assert np.array_equal(_decode(a), img)
    # the model predicts the same bytes at full size for a band of rows (the whole frame takes the Python model too long)
    rows = [0, 1, 150, 350, 719]
    for y in rows:
        seg = P.encode_row(P.filter_row(img[y]))
        assert seg in a, f"row {y}"


def test_png_widest_supported_row_and_the_limit(favlib, oracle, cuda):
    """a row lives in one CU's LDS (raw + filtered + token descriptors + bit buffer): 9000 pixels is the limit the entry points state;
    the widest case runs with the raised dynamic-LDS attribute and decodes exactly, one pixel more is refused with a status"""
    import png_model as P
    rng = np.random.default_rng(4)
    img = np.ascontiguousarray(np.repeat(rng.integers(0, 256, (2, 1800, 3), dtype=np.uint8), 5, axis=1))      # 9000 wide, runs of 5 pixels
    img[1, 4000:4600] = rng.integers(0, 256, (600, 3), dtype=np.uint8)
    data = favlib.png_encode(T(img, cuda))
    assert np.array_equal(_decode(data), img) and len(data) <= P.capacity(9000, 2)
    assert data == P.encode(img)
    with pytest.raises(favlib.FavError, match="9000"):
        favlib.png_encode(T(np.zeros((1, 9001, 3), np.uint8), cuda))


def test_png_above_16_mib_has_a_valid_crc(favlib, oracle, cuda):
    """noise at 3000x2000 is stored almost raw -- an 18 MB IDAT chunk, beyond the 2^24 bytes the CRC combine's three
    position tables covered (PIL / libpng verify the chunk CRC and rejected such files); also an unpadded, unaligned source buffer"""
    import torch
    h, w = 2000, 3000
    rng = np.random.default_rng(11)
    img = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    data = favlib.png_encode(T(img, cuda))
    assert len(data) > (1 << 24)
    import struct, zlib
    assert data[33:37] == struct.pack(">I", len(data) - 57) and data[37:41] == b"IDAT"
    assert struct.unpack(">I", data[-16:-12])[0] == zlib.crc32(data[37:-16])            # the IDAT chunk's CRC over type + data
    assert np.array_equal(_decode(data), img)
    # a source that starts at an odd address and ends at the last byte of its allocation: nothing outside it is read, same file
    hs, ws = 37, 333
    small = np.ascontiguousarray(img[:hs, :ws])
    buf = torch.zeros(1 + small.size, dtype=torch.uint8, device=cuda)
    buf[1:] = T(small.reshape(-1), cuda)
    assert favlib.png_encode(buf[1:].view(hs, ws, 3)) == favlib.png_encode(T(small, cuda))
