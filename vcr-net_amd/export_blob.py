"""Packed weights + one batch of clouds as a flat file for hosts without Python (examples/host_cpp/forward_host.cpp):
the image of a module's `vcr_vcrnet_weights` with its pointers replaced by (struct offset -> blob offset) patches, and the
blob of the packed device tensors those pointers meant.  Test / demonstration plumbing around the C-ABI; no arithmetic."""
from __future__ import annotations

import ctypes as C
import struct

import torch

from . import native

MAGIC = 0x42524356          # 'VCRB'


def _pointer_fields(ct, base=0):
    """(byte offset, value) of every pointer-typed field of a ctypes struct instance, nested structs included."""
    out = []
    for name, typ in ct._fields_:
        off = base + getattr(type(ct), name).offset
        val = getattr(ct, name)
        if isinstance(val, C.Structure):
            out += _pointer_fields(val, off)
        elif typ is C.c_void_p:
            out.append((off, val or 0))
    return out


def write_blob(net, src: torch.Tensor, tgt: torch.Tensor, path: str) -> None:
    """net: a vcrnet_amd.module.VCRNet on a GPU whose weights are packed (one forward has run, or net._pack() was called)."""
    with torch.cuda.device(src.device):
        net._pack()
    cw, P = net._cw, net._packed
    tensors = []
    for v in P.values():
        tensors += [t for t in (v if isinstance(v, (tuple, list)) else (v,)) if torch.is_tensor(t)]
    spans = sorted({(t.data_ptr(), t.numel() * t.element_size()): t for t in tensors}.items())
    blob, where = bytearray(), []
    for (ptr, nbytes), t in spans:
        blob += b"\0" * ((-len(blob)) % 256)
        where.append((ptr, nbytes, len(blob)))
        blob += t.detach().contiguous().cpu().view(torch.uint8).numpy().tobytes() if t.dtype != torch.uint8 else t.cpu().numpy().tobytes()
    patches = []
    for off, val in _pointer_fields(cw):
        if not val:
            continue
        hit = [(p, n, b) for p, n, b in where if p <= val < p + n]
        if not hit:
            raise native.VcrHipError(f"weights field at struct offset {off} points outside the packed tensors")
        p, _, b = hit[0]
        patches.append((off, b + (val - p)))
    clouds = []
    for x in (src, tgt):
        blob += b"\0" * ((-len(blob)) % 256)
        clouds.append(len(blob))
        blob += x.detach().contiguous().float().cpu().numpy().tobytes()
    image = bytearray(bytes(cw))
    for off, _ in patches:
        image[off:off + 8] = b"\0" * 8
    B, _, N = src.shape
    with open(path, "wb") as f:
        f.write(struct.pack("<6I3Q", MAGIC, native.ABI_VERSION, C.sizeof(cw), len(patches), B, N, len(blob), clouds[0], clouds[1]))
        for off, boff in patches:
            f.write(struct.pack("<IIQ", off, 0, boff))
        f.write(bytes(image))
        f.write(bytes(blob))
