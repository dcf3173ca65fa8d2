"""The reference's checkpoint container (burn 0.18 CompactRecorder, src/main.rs:839-864).  Parity unpinned: no
file written by the reference exists; the reader is checked against MessagePack assembled by hand here (so a
mistake shared by this package's reader and writer still shows) and against the package's own writer."""
import struct

import numpy as np
import pytest
import torch

import diffusionmodel_amd as A
from diffusionmodel_amd import mpk


def _mp_str(s):
    b = s.encode()
    assert len(b) < 32
    return bytes([0xA0 | len(b)]) + b


def _mp_bin(b):
    return b"\xc4" + bytes([len(b)]) + b if len(b) < 256 else b"\xc5" + struct.pack(">H", len(b)) + b


def _mp_map(pairs):
    assert len(pairs) < 16
    return bytes([0x80 | len(pairs)]) + b"".join(_mp_str(k) + v for k, v in pairs)


def _mp_arr(ints):
    return bytes([0x90 | len(ints)]) + bytes(ints)


def _mp_tensor(values, shape, dtype="F16", np_dtype="<f2"):
    raw = np.asarray(values, dtype=np_dtype).tobytes()
    return _mp_map([("id", _mp_str("7")),
                    ("param", _mp_map([("bytes", _mp_bin(raw)), ("shape", _mp_arr(shape)), ("dtype", _mp_str(dtype))]))])


def test_reader_on_hand_assembled_record(tmp_path):
    """Bytes written per the MessagePack spec, field by field, in the layout of a burn module record: nested maps
    of fields, nil for constants, {id, param{bytes, shape, dtype}} for tensors."""
    w = [[1.0, -2.5, 0.125], [3.0, 65504.0, -0.0]]
    lin = _mp_map([("weight", _mp_tensor(w, [2, 3])), ("bias", _mp_tensor([0.5, -1.0, 2.0], [3]))])
    bn = _mp_map([("gamma", _mp_tensor([1.0, 2.0], [2], "F32", "<f4")), ("epsilon", b"\xc0"), ("momentum", b"\xc0")])
    item = _mp_map([("unet", _mp_map([("time", lin), ("bn1", bn), ("pool1", b"\xc0")])), ("num_time", b"\xc0")])
    meta = _mp_map([("float", _mp_str("f16")), ("version", _mp_str("0.18.0"))])
    f = tmp_path / "m.mpk"
    f.write_bytes(_mp_map([("metadata", meta), ("item", item)]))
    t, m = mpk.read_mpk(str(f))
    assert m == {"float": "f16", "version": "0.18.0"}
    assert sorted(t) == ["unet.bn1.gamma", "unet.time.bias", "unet.time.weight"]
    assert t["unet.time.weight"].dtype == np.float32 and t["unet.time.weight"].shape == (2, 3)
    np.testing.assert_array_equal(t["unet.time.weight"], np.asarray(w, np.float32))
    assert np.signbit(t["unet.time.weight"][1, 2])
    np.testing.assert_array_equal(t["unet.time.bias"], [0.5, -1.0, 2.0])
    np.testing.assert_array_equal(t["unet.bn1.gamma"], [1.0, 2.0])
    # the same tree without the metadata wrapper
    f.write_bytes(item)
    t2, m2 = mpk.read_mpk(str(f))
    assert m2 is None and sorted(t2) == sorted(t)


def test_bf16_and_integer_array_bytes(tmp_path):
    import msgpack
    vals = np.asarray([1.0, -3.140625, 2.0 ** -20], np.float32)
    hi = (vals.view(np.uint32) >> 16).astype("<u2")
    tree = {"a": {"id": "0", "param": {"bytes": list(hi.tobytes()), "shape": [3], "dtype": "BF16"}}}
    f = tmp_path / "b.mpk"
    f.write_bytes(msgpack.packb(tree))
    t, _ = mpk.read_mpk(str(f))
    np.testing.assert_array_equal(t["a"], (hi.astype(np.uint32) << 16).view(np.float32))


@pytest.mark.parametrize("dtype,exact", [("F16", False), ("F32", True), ("BF16", False)])
def test_save_then_load_round_trip(tmp_path, dtype, exact):
    cfg = A.LDMConfig()
    w = A.make_synthetic_weights(cfg, 1)
    d = A.synthetic_from_shapes(A.decoder_param_shapes(3, 64), 2)
    f = str(tmp_path / "model.mpk")
    A.save_model_weights(f, w, d, dtype=dtype, allow_partial=True)
    u2, d2 = A.load_model_weights(f, in_channels=3, latent_dimen=64, time_embed_dim=256)
    assert list(u2) == list(A.unet_param_shapes(cfg)) and list(d2) == list(A.decoder_param_shapes(3, 64))
    for src, got in ((w, u2), (d, d2)):
        for k, v in src.items():
            if exact:
                assert torch.equal(got[k], v.float())
            elif dtype == "F16":
                assert torch.equal(got[k], v.half().float())        # exactly the half-precision rounding
            else:
                assert torch.equal(got[k], v.bfloat16().float())
    # a second save of what was loaded is byte-identical: the container is canonical for its contents
    g = str(tmp_path / "again.mpk")
    A.save_model_weights(g, u2, d2, dtype=dtype, allow_partial=True)
    assert open(f, "rb").read() == open(g, "rb").read()


def test_load_rejects_wrong_architecture_and_garbage(tmp_path):
    w = A.make_synthetic_weights(A.LDMConfig(), 1)
    f = str(tmp_path / "m.mpk")
    A.save_model_weights(f, w, allow_partial=True)
    with pytest.raises(A.MpkFormatError, match="shape"):
        A.load_model_weights(f, channels=(128, 256, 512), scaled_attention=False)
    with pytest.raises(A.MpkFormatError, match="missing"):
        A.load_model_weights(f, channels=(64, 128, 256), scaled_attention=True)
    u, d = A.load_model_weights(f)
    assert d is None                                   # a file without the decoder loads the UNet alone
    (tmp_path / "junk.mpk").write_bytes(b"\xc1\xc1\xc1")
    with pytest.raises(A.MpkFormatError):
        A.read_mpk(str(tmp_path / "junk.mpk"))
    short = {"a": {"id": "0", "param": {"bytes": b"\x00\x00", "shape": [2], "dtype": "F16"}}}
    import msgpack
    (tmp_path / "short.mpk").write_bytes(msgpack.packb(short))
    with pytest.raises(A.MpkFormatError, match="bytes for shape"):
        A.read_mpk(str(tmp_path / "short.mpk"))


@pytest.mark.gpu
def test_loaded_model_samples_like_the_model_it_was_saved_from(tmp_path):
    """F32 container: the loaded model is the same model, so its chain is bit-identical; F16 container (what the
    reference writes): the chain equals that of a model built from the half-rounded weights."""
    w = A.make_synthetic_weights(A.LDMConfig(), 5)
    d = A.synthetic_from_shapes(A.decoder_param_shapes(3, 64), 6)
    z0 = torch.randn(2, 64, 4, 4, generator=torch.Generator().manual_seed(0)).cuda()
    base = A.DiffusionModel(64, 3, 50, 256, dtype="bf16", weights=w, decoder_weights=d)
    ref = base.decode(base.ddim_sample(2, 5, spatial=4, z0=z0))
    f = str(tmp_path / "m.mpk")
    A.save_model_weights(f, w, d, dtype="F32", allow_partial=True)
    m = A.load_model(f, 3, 64, 50, 256)
    assert torch.equal(m.decode(m.ddim_sample(2, 5, spatial=4, z0=z0)), ref)
    A.save_model_weights(f, w, d, dtype="F16", allow_partial=True)
    m16 = A.load_model(f, 3, 64, 50, 256)
    half = lambda t: {k: v.half().float() for k, v in t.items()}
    b16 = A.DiffusionModel(64, 3, 50, 256, dtype="bf16", weights=half(w), decoder_weights=half(d))
    assert torch.equal(m16.decode(m16.ddim_sample(2, 5, spatial=4, z0=z0)), b16.decode(b16.ddim_sample(2, 5, spatial=4, z0=z0)))


def test_full_record_has_every_field_of_every_struct(tmp_path):
    """What burn's load_record needs (src/main.rs:860-861): every field of DiffusionModel, Unet, Vae, SimpleEncoder,
    SimpleDecoder and of the burn layers below them present as a key: tensors as {"id", "param"} maps, constant
    fields as nil.  The written tree is walked against the struct inventory (src/main.rs:22-33, 96-106, 172-175,
    238-243, 279-282, 337-343, 390-414, 530-535)."""
    import msgpack
    from diffusionmodel_amd.weights import encoder_param_shapes
    cfg = A.LDMConfig()
    w = A.make_synthetic_weights(cfg, 1)
    d = A.synthetic_from_shapes(A.decoder_param_shapes(3, 64), 2)
    e = A.synthetic_from_shapes(encoder_param_shapes(3, 64), 3)
    f = str(tmp_path / "full.mpk")
    with pytest.raises(ValueError, match="missing"):
        A.save_model_weights(f, w, d)                      # no encoder: not a record load_model could take
    A.save_model_weights(f, w, d, encoder=e)
    tree = msgpack.unpackb(open(f, "rb").read(), raw=False)
    assert set(tree) == {"metadata", "item"} and tree["metadata"]["float"] == "f16"
    item = tree["item"]
    assert set(item) == {"unet", "vae", "num_time", "latent_dimen"} and item["num_time"] is None
    assert set(item["vae"]) == {"encoder", "decoder"}
    unet_fields = {"down1", "down_res1", "pool1", "down2", "down_res2", "pool2", "down3", "down_res3", "mid_res1", "mid_res2",
                   "mid_atten", "mid_drop", "up1", "up_conv1", "up_res1", "up2", "up_conv2", "up_res2", "out_conv", "time_emb"}
    assert set(item["unet"]) == unet_fields
    assert item["unet"]["pool1"] is None and item["unet"]["mid_drop"] is None
    assert set(item["unet"]["down1"]) == {"weight", "bias", "stride", "kernel_size", "dilation", "groups", "padding"}
    assert set(item["unet"]["up1"]) == {"weight", "bias", "stride", "kernel_size", "dilation", "groups", "padding", "padding_out", "channels"}
    assert set(item["unet"]["down_res1"]) == {"conv1", "conv2", "bn1", "bn2", "time"}
    assert set(item["unet"]["down_res1"]["bn1"]) == {"gamma", "beta", "running_mean", "running_var", "momentum", "epsilon"}
    assert set(item["unet"]["down_res1"]["time"]) == {"weight", "bias"}
    assert set(item["unet"]["mid_atten"]) == {"question", "key", "val", "out", "channels"}
    assert set(item["unet"]["time_emb"]) == {"linear1", "linear2"}
    assert set(item["vae"]["encoder"]) == {"conv1", "bn_1", "conv2", "bn_2", "conv3", "bn_3", "mean", "variance", "drop", "latent_dimen"}
    assert set(item["vae"]["decoder"]) == {"transfer", "drop1", "reverse1", "bn_1", "reverse2", "bn_2", "reverse3", "latent_dimen", "output"}
    # every leaf is a tensor of the inventory's shape or nil, and nothing else is in the tree
    inv = mpk.record_inventory(3, 64, 256)
    seen = {}

    def walk(node, path):
        if node is None or mpk._is_tensor(node):
            seen[path] = None if node is None else tuple(node["param"]["shape"])
        else:
            assert isinstance(node, dict), path
            for k, v in node.items():
                walk(v, f"{path}.{k}" if path else k)
    walk(item, "")
    assert seen == inv
    # and this package reads its own full record back
    u2, d2 = A.load_model_weights(f)
    assert torch.equal(u2["down1.weight"], w["down1.weight"].half().float()) and d2 is not None
