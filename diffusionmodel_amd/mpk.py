"""Weights in and out of the reference's checkpoint container.

The reference saves a DiffusionModel with burn 0.18's `CompactRecorder` (src/main.rs:839-847) and would load it
with the same recorder (:849-864): a named MessagePack file (rmp-serde 1.3 `to_vec_named`) of

    {"metadata": {"float": "f16", "int": "i16", "format": ..., "version": "0.18.0", "settings": ...},
     "item": {"unet": {...}, "vae": {"encoder": {...}, "decoder": {...}}, "num_time": nil, "latent_dimen": nil}}

where every module is a map of its fields (src/main.rs:22-33, 96-106, 238-243, 279-282, 337-343, 390-414,
530-535), every parameter or running statistic is `{"id": str, "param": {"bytes": bin, "shape": [..],
"dtype": "F16"}}` with little-endian elements in row-major order, and every non-parameter field (usize, Dropout,
MaxPool2d, padding configs) is nil.  Linear weights are [d_in, d_out], Conv2d [out, in, kh, kw],
ConvTranspose2d [in, out, kh, kw]: the layouts `unet_param_shapes` / `decoder_param_shapes` already use, so the
dotted path of a tensor below "unet." or "vae.decoder." is its name here.

PARITY UNPINNED: no checkpoint ships with the reference (its .gitignore excludes *.mpk) and its Rust cannot be
built here, so this layout follows burn 0.18's published record format and is checked only against files this
module writes and against hand-assembled MessagePack in the tests.  The reader is therefore tolerant: it takes the
tree with or without the metadata wrapper, bytes as bin or as an integer array, and F16 / BF16 / F32 / F64
elements.
"""
from __future__ import annotations

import os
from typing import Dict, Optional, Tuple

import msgpack
import numpy as np

_DTYPES = {"F16": ("<f2", 2), "F32": ("<f4", 4), "F64": ("<f8", 8), "BF16": (None, 2)}
BURN_VERSION = "0.18.0"
UNET_PREFIX = "unet."
DECODER_PREFIX = "vae.decoder."


class MpkFormatError(ValueError):
    """The file is not a named-MessagePack module record of the layout described in this module."""


def _is_tensor(node) -> bool:
    return isinstance(node, dict) and "param" in node and isinstance(node["param"], dict) \
        and "bytes" in node["param"] and "shape" in node["param"]


def _decode_tensor(path: str, node) -> np.ndarray:
    rec = node["param"]
    dtype = rec.get("dtype", "F16")
    if isinstance(dtype, dict):      # a data-carrying enum variant (quantised tensors) arrives as a one-key map
        raise MpkFormatError(f"{path}: quantised dtype {list(dtype)} is not supported")
    if dtype not in _DTYPES:
        raise MpkFormatError(f"{path}: dtype {dtype!r} is not a float type")
    raw = rec["bytes"]
    if isinstance(raw, (list, tuple)):
        raw = bytes(bytearray(raw))
    if not isinstance(raw, (bytes, bytearray, memoryview)):
        raise MpkFormatError(f"{path}: bytes field has type {type(raw).__name__}")
    shape = tuple(int(d) for d in rec["shape"])
    np_dtype, width = _DTYPES[dtype]
    count = int(np.prod(shape, dtype=np.int64)) if shape else 1
    if len(raw) != count * width:
        raise MpkFormatError(f"{path}: {len(raw)} bytes for shape {shape} of {dtype}")
    if dtype == "BF16":
        wide = np.frombuffer(raw, dtype="<u2").astype(np.uint32) << 16
        arr = wide.view(np.float32)
    else:
        arr = np.frombuffer(raw, dtype=np_dtype).astype(np.float32)
    return np.ascontiguousarray(arr.reshape(shape))


def _walk(node, path: str, out: Dict[str, np.ndarray]) -> None:
    if _is_tensor(node):
        out[path] = _decode_tensor(path, node)
    elif isinstance(node, dict):
        for key, child in node.items():
            _walk(child, f"{path}.{key}" if path else str(key), out)
    elif isinstance(node, (list, tuple)):     # Vec<Module> fields record as arrays
        for i, child in enumerate(node):
            _walk(child, f"{path}.{i}" if path else str(i), out)
    # nil and scalars: constants of the module, nothing to load


def read_mpk(path: str) -> Tuple[Dict[str, np.ndarray], Optional[dict]]:
    """Every tensor of the record as float32, keyed by its dotted field path, plus the metadata map (or None)."""
    with open(path, "rb") as fh:
        try:
            tree = msgpack.unpackb(fh.read(), raw=False, strict_map_key=False)
        except Exception as exc:   # msgpack raises several unrelated types
            raise MpkFormatError(f"{path}: not MessagePack ({exc})") from exc
    if not isinstance(tree, dict):
        raise MpkFormatError(f"{path}: top level is {type(tree).__name__}, expected a map")
    meta = tree.get("metadata") if "item" in tree else None
    item = tree["item"] if "item" in tree else tree
    out: Dict[str, np.ndarray] = {}
    _walk(item, "", out)
    if not out:
        raise MpkFormatError(f"{path}: no tensors found")
    return out, meta


def _strip(tensors: Dict[str, np.ndarray], prefix: str) -> Dict[str, np.ndarray]:
    return {k[len(prefix):]: v for k, v in tensors.items() if k.startswith(prefix)}


def _check(got: Dict[str, np.ndarray], shapes: Dict[str, tuple], what: str) -> Dict[str, np.ndarray]:
    missing = [k for k in shapes if k not in got]
    if missing:
        raise MpkFormatError(f"{what}: {len(missing)} tensors missing, first {missing[0]!r}")
    for k, shp in shapes.items():
        if tuple(got[k].shape) != tuple(shp):
            raise MpkFormatError(f"{what}: {k} has shape {tuple(got[k].shape)}, the architecture needs {tuple(shp)}")
    return {k: got[k] for k in shapes}


def load_model_weights(path: str, in_channels: int = 3, latent_dimen: int = 64, time_embed_dim: int = 256,
                       channels=(64, 128, 256), scaled_attention: bool = False):
    """The two weight dictionaries `DiffusionModel(weights=..., decoder_weights=...)` takes, from a file written by
    the reference's save_model (src/main.rs:839-847).  Shapes are checked against the architecture the arguments
    name, as burn's load_record would (:860-861).  The VAE encoder's tensors are read and dropped."""
    from .weights import LDMConfig, decoder_param_shapes, unet_param_shapes
    tensors, _ = read_mpk(path)
    if not any(k.startswith(UNET_PREFIX) for k in tensors):
        # a bare Unet record (someone saved model.unet alone): accept it as the UNet
        unet_t, dec_t = tensors, {}
    else:
        unet_t, dec_t = _strip(tensors, UNET_PREFIX), _strip(tensors, DECODER_PREFIX)
    cfg = LDMConfig(latent_dimen, time_embed_dim, tuple(channels), scaled_attention)
    import torch
    unet = _check(unet_t, unet_param_shapes(cfg), "unet")
    decoder = _check(dec_t, decoder_param_shapes(in_channels, latent_dimen), "vae.decoder") if dec_t else None
    as_torch = lambda d: None if d is None else {k: torch.from_numpy(v) for k, v in d.items()}
    return as_torch(unet), as_torch(decoder)


def load_model(path: str, in_channels: int, latent_dimen: int, num_time: int, time_embed_dim: int, device="cuda",
               dtype="bf16", channels=(64, 128, 256), scaled_attention: bool = False):
    """load_model(path, in_channels, latent_dimen, num_time, time_embed_dim, device) of the reference
    (src/main.rs:849-864): build the model for those arguments and load the record at `path` into it."""
    from .modules import DiffusionModel
    unet, decoder = load_model_weights(path, in_channels, latent_dimen, time_embed_dim, channels, scaled_attention)
    return DiffusionModel(latent_dimen, in_channels, num_time, time_embed_dim, channels=channels,
                          scaled_attention=scaled_attention, dtype=dtype, weights=unet, device=device,
                          decoder_weights=decoder)


def _encode_tensor(arr: np.ndarray, name: str, dtype: str) -> dict:
    a = np.ascontiguousarray(np.asarray(arr, dtype=np.float32))
    if dtype == "BF16":
        bits = a.view(np.uint32)
        rounded = ((bits + 0x7FFF + ((bits >> 16) & 1)) >> 16).astype("<u2")   # round to nearest even
        raw = rounded.tobytes()
    else:
        raw = a.astype(_DTYPES[dtype][0]).tobytes()
    return {"id": name, "param": {"bytes": raw, "shape": [int(d) for d in a.shape], "dtype": dtype}}


def _insert(tree: dict, dotted: str, leaf: dict) -> None:
    parts = dotted.split(".")
    for p in parts[:-1]:
        tree = tree.setdefault(p, {})
    tree[parts[-1]] = leaf


# Fields of burn 0.18's layer modules that are not parameters: `#[derive(Module)]` records each as a constant, which
# the named MessagePack recorder writes as nil, and `load_record` wants every field of the struct present as a key.
_CONV_CONSTS = ("stride", "kernel_size", "dilation", "groups", "padding")
_CONVT_CONSTS = ("stride", "kernel_size", "dilation", "groups", "padding", "padding_out", "channels")
_BN_CONSTS = ("momentum", "epsilon")
ENCODER_PREFIX = "vae.encoder."


def record_inventory(in_channels: int = 3, latent_dimen: int = 64, time_embed_dim: int = 256, channels=(64, 128, 256),
                     scaled_attention: bool = False) -> Dict[str, Optional[tuple]]:
    """Every leaf of the DiffusionModel record by dotted field path: the shape of a tensor, or None for a constant
    field (written as nil).  Struct fields from src/main.rs:22-33 (SimpleEncoder), 96-106 (SimpleDecoder), 172-175
    (Vae), 238-243 (Attention), 279-282 (TimeAddition), 337-343 (ResidualBlock), 390-414 (Unet), 530-535
    (DiffusionModel); the fields of Conv2d / ConvTranspose2d / BatchNorm / Linear from burn 0.18's layer structs."""
    from .weights import LDMConfig, decoder_param_shapes, encoder_param_shapes, unet_param_shapes
    inv: Dict[str, Optional[tuple]] = {}

    def layer_consts(prefix: str, tensors: Dict[str, tuple], convT: set):
        layers = {}
        for name in tensors:
            layer, leaf = name.rsplit(".", 1)
            layers.setdefault(layer, set()).add(leaf)
        for layer, leaves in layers.items():
            if "gamma" in leaves:
                consts = _BN_CONSTS
            elif layer.split(".")[-1] in convT:
                consts = _CONVT_CONSTS
            elif len(tensors[layer + ".weight"]) == 4:
                consts = _CONV_CONSTS
            else:
                consts = ()              # Linear: weight and bias only
            for c in consts:
                inv[prefix + layer + "." + c] = None

    cfg = LDMConfig(latent_dimen, time_embed_dim, tuple(channels), scaled_attention)
    unet = unet_param_shapes(cfg)
    for k, shp in unet.items():
        inv[UNET_PREFIX + k] = tuple(shp)
    layer_consts(UNET_PREFIX, unet, {"up1", "up2"})
    for c in ("pool1", "pool2", "mid_drop", "mid_atten.channels"):
        inv[UNET_PREFIX + c] = None
    if scaled_attention:
        inv[UNET_PREFIX + "up_atten.channels"] = None
    dec = decoder_param_shapes(in_channels, latent_dimen)
    for k, shp in dec.items():
        inv[DECODER_PREFIX + k] = tuple(shp)
    layer_consts(DECODER_PREFIX, dec, {"reverse1", "reverse2", "reverse3"})
    for c in ("drop1", "latent_dimen", "output"):
        inv[DECODER_PREFIX + c] = None
    enc = encoder_param_shapes(in_channels, latent_dimen)
    for k, shp in enc.items():
        inv[ENCODER_PREFIX + k] = tuple(shp)
    layer_consts(ENCODER_PREFIX, enc, set())
    for c in ("drop", "latent_dimen"):
        inv[ENCODER_PREFIX + c] = None
    inv["num_time"] = None
    inv["latent_dimen"] = None
    return inv


def save_model_weights(path: str, unet: Dict[str, np.ndarray], decoder: Optional[Dict[str, np.ndarray]] = None,
                       dtype: str = "F16", extra: Optional[Dict[str, np.ndarray]] = None,
                       encoder: Optional[Dict[str, np.ndarray]] = None, in_channels: int = 3, latent_dimen: int = 64,
                       time_embed_dim: int = 256, channels=(64, 128, 256), scaled_attention: bool = False,
                       allow_partial: bool = False) -> None:
    """Write the record save_model writes (src/main.rs:839-847): half-precision elements by default, every field of
    every struct present: tensors as {"id", "param"}, constant fields (strides, paddings, Dropout, MaxPool2d, usize)
    as nil, following `record_inventory`.  burn's load_record (:860-861) wants the whole DiffusionModel, so the
    decoder and the VAE encoder's tensors (`encoder`, names as in weights.encoder_param_shapes) are required;
    allow_partial=True writes whatever is given (a file this package's own reader takes, but NOT one the
    reference's load_model can load).  `extra` takes further full paths.
    PARITY UNPINNED: whether burn accepts the file could not be tried here (no cargo, no checkpoint of the reference)."""
    if dtype not in _DTYPES:
        raise ValueError(f"dtype {dtype!r}")
    to_np = lambda v: v.detach().cpu().numpy() if hasattr(v, "detach") else v
    given: Dict[str, np.ndarray] = {}
    for prefix, group in ((UNET_PREFIX, unet), (DECODER_PREFIX, decoder), (ENCODER_PREFIX, encoder), ("", extra)):
        for k, v in (group or {}).items():
            given[prefix + k] = np.asarray(to_np(v))
    inv = record_inventory(in_channels, latent_dimen, time_embed_dim, channels, scaled_attention)
    missing = [k for k, shp in inv.items() if shp is not None and k not in given]
    if missing and not allow_partial:
        raise ValueError(f"{len(missing)} tensors of the DiffusionModel record are missing (first {missing[0]!r}); the "
                         "reference's load_model needs all of them (pass the decoder and the encoder, or allow_partial=True "
                         "for a file only this package reads)")
    for k, shp in inv.items():
        if shp is not None and k in given and tuple(given[k].shape) != tuple(shp):
            raise ValueError(f"{k} has shape {tuple(given[k].shape)}, the architecture needs {tuple(shp)}")
    item: dict = {}
    for k, v in given.items():
        _insert(item, k, _encode_tensor(v, k, dtype))
    missing_set = set(missing)
    for k, shp in inv.items():
        if shp is None:
            # a constant below a module none of whose tensors was given (allow_partial) would only create an empty shell
            top = k.rsplit(".", 1)[0] + "." if "." in k else ""
            if not allow_partial or not top or any(g.startswith(top) for g in given) or k in ("num_time", "latent_dimen"):
                _insert(item, k, None)
    meta = {"float": {"F16": "f16", "BF16": "bf16", "F32": "f32", "F64": "f64"}[dtype], "int": "i16",
            "format": "burn_core::record::file::NamedMpkFileRecorder", "version": BURN_VERSION,
            # burn writes core::any::type_name of the settings type here (a full path) and only ever reads it back
            # for the text of a mismatch error
            "settings": "burn_core::record::settings::HalfPrecisionSettings" if dtype == "F16" else "custom"}
    tmp = path + ".tmp"
    with open(tmp, "wb") as fh:
        fh.write(msgpack.packb({"metadata": meta, "item": item}, use_bin_type=True))
    os.replace(tmp, path)
