"""Packed on-disk form of a GPFQ-quantized network (new in this build; the reference saves the fp32 model with
torch.save, main.py:127-136, which is 32 bits per weight for an alphabet of 2K+1 values).

Per quantized layer the file keeps the alphabet INDICES the loop kernel emits next to Q -- n-bit fields,
n = ceil(log2(number of alphabet values)), eight fields per n bytes -- plus the fp32 alphabet step, and the weights are
rebuilt with the kernel's own fp32 operations (dist.rebuild_q: (sign*step)*|k| for msq / soft / stochastic,
step_algorithm.py:56; sign*(lamb + step*k) for hard, :81), so a load is BITWISE the quantized model.
Everything that was not quantized (biases, normalisation layers, ignored layers) is stored as it is.

    from quantized_neural_nets_amd import packed
    packed.save(path, quantizer)                    # after quantizer.quantize_network()
    model = packed.load(path, model_with_the_same_architecture)
"""
import math

import torch

from . import dist as _dist
from .utils import extract_layers

FORMAT = "gpfq-packed-v1"
MODE_HARD = 2


def index_range(K, mode):
    """(lowest index, number of index values) of the alphabet: msq / soft / stochastic -K..K; hard 0, +-1..+-(K+1)."""
    hi = K + 1 if mode == MODE_HARD else K
    return -hi, 2 * hi + 1


def field_bits(K, mode):
    return max(1, math.ceil(math.log2(index_range(K, mode)[1])))


def pack_indices(idx, K, mode):
    """idx: integer tensor of alphabet indices (any shape) -> (uint8 tensor, nbits).  Eight n-bit fields, lowest
    index first, little-endian inside n bytes; the tail is padded with the lowest index."""
    lo, count = index_range(K, mode)
    nbits = field_bits(K, mode)
    v = idx.reshape(-1).to(torch.int64) - lo
    if v.numel() and (int(v.min()) < 0 or int(v.max()) >= count):
        raise ValueError("index outside the alphabet")
    if nbits > 7:                                    # wide alphabets are stored as they are
        return idx.reshape(-1).to(torch.int16).contiguous().view(torch.uint8), 16
    pad = (-v.numel()) % 8
    if pad:
        v = torch.cat([v, torch.zeros(pad, dtype=torch.int64, device=v.device)])
    v = v.view(-1, 8)
    shifts = torch.arange(8, device=v.device, dtype=torch.int64) * nbits
    word = (v << shifts).sum(1)                      # < 2^56
    byte_shifts = torch.arange(nbits, device=v.device, dtype=torch.int64) * 8
    return ((word[:, None] >> byte_shifts) & 0xFF).to(torch.uint8).reshape(-1), nbits


def unpack_indices(packed_bytes, nbits, numel, K, mode):
    lo, _ = index_range(K, mode)
    if nbits == 16:
        return packed_bytes.view(torch.int16)[:numel].to(torch.int64)
    b = packed_bytes.view(-1, nbits).to(torch.int64)
    byte_shifts = torch.arange(nbits, device=b.device, dtype=torch.int64) * 8
    word = (b << byte_shifts).sum(1)
    shifts = torch.arange(8, device=b.device, dtype=torch.int64) * nbits
    v = (word[:, None] >> shifts) & ((1 << nbits) - 1)
    return v.reshape(-1)[:numel] + lo


def save(path, quantizer):
    """Write the quantized network held by a QuantizeNeuralNet after quantize_network()."""
    layers = {}
    packed_names = set()
    names = {id(m): n for n, m in quantizer.quantized_network.named_modules()}
    for rec in quantizer.layer_indices:
        layer = quantizer.quantized_network_layers[rec["layer"]]
        data, nbits = pack_indices(rec["idx"].cpu(), rec["K"], rec["mode"])
        name = names[id(layer)]
        layers[name] = dict(packed=data, nbits=nbits, shape=tuple(layer.weight.shape), step=float(rec["step"]),
                            K=int(rec["K"]), mode=int(rec["mode"]), lamb=float(rec["lamb"]))
        packed_names.add(name + ".weight")
    rest = {k: v.detach().cpu() for k, v in quantizer.quantized_network.state_dict().items() if k not in packed_names}
    torch.save(dict(format=FORMAT, layers=layers, state_dict=rest), path)
    dense = sum(math.prod(v["shape"]) * 4 for v in layers.values())
    return dict(layers=len(layers), packed_bytes=sum(v["packed"].numel() for v in layers.values()), fp32_bytes=dense)


def load(path, model, device=None):
    """Rebuild the quantized weights into `model` (same architecture); returns it."""
    blob = torch.load(path, map_location="cpu", weights_only=True)
    if blob.get("format") != FORMAT:
        raise ValueError("not a %s file" % FORMAT)
    state = dict(blob["state_dict"])
    for name, rec in blob["layers"].items():
        numel = math.prod(rec["shape"])
        idx = unpack_indices(rec["packed"], rec["nbits"], numel, rec["K"], rec["mode"])
        q = _dist.rebuild_q(idx, rec["step"], rec["K"], rec["mode"], rec["lamb"])
        state[name + ".weight"] = q.reshape(rec["shape"])
    model.load_state_dict(state)
    return model.to(device) if device is not None else model
