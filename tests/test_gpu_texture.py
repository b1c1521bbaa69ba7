"""The DDS / BC7 texture path on the GPU (reference entry rt64_texture.cpp:146-187, where the blocks go to the sampler hardware undecoded):
bc7_decode_kernel (csrc/bc7.hip) decodes every block once at RT64_CreateTexture.  Its texels -- read back with RT64_ReadbackTexture -- against the
oracle's decoder (oracle/oracle_texture.c, itself pinned against Pillow's on every mode in tests/test_oracle_texture.py): every mode, every partition of
the partitioned modes, every rotation / index-selection of modes 4 and 5, and a mip chain whose small levels are partial blocks.  Bit-exact (bytes)."""
import ctypes as C
import struct

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

# (mode, partition bits): BC7 modes 0, 1, 2, 3, 7 carry a partition index right behind the mode bits
PARTITION_BITS = {0: 4, 1: 6, 2: 6, 3: 6, 7: 6}


def _dds_bc7(width, height, mips, raw):
    hdr = struct.pack("<4s7I44xII4s5I4I4x", b"DDS ", 124, 0x1007 | 0x80000 | (0x20000 if mips > 1 else 0), height, width, len(raw), 0, mips, 32, 4, b"DX10", 0, 0, 0, 0, 0, 0x1000, 0, 0, 0)
    return np.frombuffer(hdr + struct.pack("<5I", 98, 3, 0, 1, 0) + raw, dtype=np.uint8).copy()


def _force(block_int, mode, field=None, field_bits=0):
    """The 128-bit block with its mode bits forced to `mode` and (optionally) the field right behind them forced to `field`."""
    v = (block_int >> (mode + 1) << (mode + 1)) | (1 << mode)
    if field_bits:
        lo = mode + 1
        v = (v & ~(((1 << field_bits) - 1) << lo)) | (field << lo)
    return v


def _mip_sizes(w, h, mips):
    out = []
    for _ in range(mips):
        out.append((w, h)); w = max(w // 2, 1); h = max(h // 2, 1)
    return out


def _both_decoders(rt64_lib, oracle_lib, sample_data, dds, width, height, mips):
    """(GPU texels, oracle texels) of every mip level of a DDS: RT64_CreateTexture + RT64_ReadbackTexture / oracle_texture_create_dds + oracle_texture_mip."""
    from sm64rt_legacy_renderer_amd import rt64
    dev = rt64_lib.CreateDeviceHeadless(64, 64, 0)
    assert dev, rt64_lib.last_error()
    gpu, ref = [], []
    try:
        d = rt64.TEXTURE_DESC()
        d.bytes = dds.ctypes.data; d.byteCount = dds.nbytes; d.format = rt64.TEXTURE_FORMAT_DDS; d.width = d.height = d.rowPitch = -1
        t = rt64_lib.CreateTexture(dev, d)
        assert t, rt64_lib.last_error()
        for m, (mw, mh) in enumerate(_mip_sizes(width, height, mips)):
            a = np.zeros((mh, mw, 4), dtype=np.uint8)
            assert rt64_lib.ReadbackTexture(t, m, None, 0) == a.nbytes
            assert rt64_lib.ReadbackTexture(t, m, a.ctypes.data, a.nbytes) == a.nbytes, rt64_lib.last_error()
            gpu.append(a)
        assert rt64_lib.ReadbackTexture(t, mips, None, 0) == 0          # no such level
        rt64_lib.DestroyTexture(t)
    finally:
        rt64_lib.DestroyDevice(dev)
    o = oracle_lib.oracle_texture_create_dds(dds.ctypes.data, dds.nbytes)
    assert o
    try:
        for m, (mw, mh) in enumerate(_mip_sizes(width, height, mips)):
            w, h = C.c_int(), C.c_int()
            p = oracle_lib.oracle_texture_mip(o, m, C.byref(w), C.byref(h))
            assert (w.value, h.value) == (mw, mh)
            ref.append(np.ctypeslib.as_array(p, shape=(mh, mw, 4)).copy())
    finally:
        oracle_lib.oracle_texture_destroy(o)
    return gpu, ref


def test_bc7_decode_kernel_every_mode_partition_and_rotation(rt64_lib, oracle_lib, sample_data):
    rng = np.random.default_rng(11)
    blocks = []

    def rnd():
        return int.from_bytes(rng.integers(0, 256, size=16, dtype=np.uint8).tobytes(), "little")
    for mode in range(8):                                   # 400 random blocks per mode (the set tests/test_oracle_texture.py holds against Pillow)
        blocks += [_force(rnd(), mode) for _ in range(400)]
    for mode, bits in PARTITION_BITS.items():               # every partition of every partitioned mode, six random fillings each
        for part in range(1 << bits):
            blocks += [_force(rnd(), mode, part, bits) for _ in range(6)]
    for rot_idx in range(8):                                # mode 4: 2 rotation bits + 1 index-selection bit
        blocks += [_force(rnd(), 4, rot_idx, 3) for _ in range(24)]
    for rot in range(4):                                    # mode 5: 2 rotation bits
        blocks += [_force(rnd(), 5, rot, 2) for _ in range(24)]
    blocks.append(0)                                        # the reserved mode (no mode bit set) decodes to zeros
    blocks.append((1 << 128) - 1)
    bw = 64
    while len(blocks) % bw:
        blocks.append(_force(rnd(), len(blocks) % 8))
    bh = len(blocks) // bw
    raw = b"".join(b.to_bytes(16, "little") for b in blocks)
    gpu, ref = _both_decoders(rt64_lib, oracle_lib, sample_data, _dds_bc7(4 * bw, 4 * bh, 1, raw), 4 * bw, 4 * bh, 1)
    bad = np.argwhere((gpu[0] != ref[0]).any(axis=-1))
    assert len(bad) == 0, (len(bad), [(int(y) // 4 * bw + int(x) // 4) for y, x in bad[:8]])
    modes = {min((k for k in range(8) if (b >> k) & 1), default=8) for b in blocks}
    assert modes == set(range(9))                           # all eight modes and the reserved one were in the texture


def test_bc7_mip_chain_with_partial_blocks(rt64_lib, oracle_lib, sample_data):
    """A 20 x 12 BC7 texture with its five levels 20x12, 10x6, 5x3, 2x1, 1x1: levels that are not multiples of four store whole blocks and show part of them
    (the sample's grass_dif.dds ends the same way)."""
    rng = np.random.default_rng(12)
    w, h, mips = 20, 12, 5
    raw = b""
    for mw, mh in _mip_sizes(w, h, mips):
        n = ((mw + 3) // 4) * ((mh + 3) // 4)
        for k in range(n):
            v = int.from_bytes(rng.integers(0, 256, size=16, dtype=np.uint8).tobytes(), "little")
            raw += _force(v, (k + mw) % 8).to_bytes(16, "little")
    gpu, ref = _both_decoders(rt64_lib, oracle_lib, sample_data, _dds_bc7(w, h, mips, raw), w, h, mips)
    for m in range(mips):
        assert np.array_equal(gpu[m], ref[m]), m


def test_sample_dds_texture_decodes_like_the_oracle(rt64_lib, oracle_lib, sample_data):
    """grass_dif.dds (512 x 512, 10 levels): every level of the texture the sample's sphere is shaded with."""
    dds = np.ascontiguousarray(sample_data.textures[0].data)
    gpu, ref = _both_decoders(rt64_lib, oracle_lib, sample_data, dds, 512, 512, 10)
    for m in range(10):
        assert np.array_equal(gpu[m], ref[m]), m
