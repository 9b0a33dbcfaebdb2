"""Writes tests/golden/rgbe_ref_vectors.npz: what the REFERENCE'S OWN Radiance reader (Common/rgbe.c, compiled unmodified
from /root/reference into oracle/_ref/librgbe_ref.so by `make -C oracle ref`) answers for every byte string of
rgbe_corpus.corpus() -- accept / reject, size, and the decoded floats -- and what its writer emits for three images.
Data only: inputs written by rgbe_corpus.py, outputs computed by the reference.  Run in the build container:
    python tests/golden/make_rgbe_ref_vectors.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)
from rgbe_corpus import corpus            # noqa: E402
from rgbe_ref import RgbeRef              # noqa: E402


def main():
    ref = RgbeRef()
    names, blobs, status, dims, pix = [], [], [], [], []
    for name, data in corpus(fuzz=300):
        st, img = ref.decode(data)
        names.append(name); blobs.append(np.frombuffer(data, np.uint8)); status.append(st)
        dims.append(img.shape[:2] if img is not None else (0, 0))
        pix.append(img.reshape(-1).view(np.uint32) if img is not None else np.zeros(0, np.uint32))
    off = np.cumsum([0] + [len(b) for b in blobs])
    poff = np.cumsum([0] + [len(p) for p in pix])
    out = os.path.join(HERE, "rgbe_ref_vectors.npz")
    np.savez_compressed(out, names=np.array(names), data=np.concatenate(blobs), offsets=off, status=np.array(status, np.int8),
                        dims=np.array(dims, np.int32), pixels=np.concatenate(pix), pixel_offsets=poff)
    print(f"{len(names)} cases, {sum(1 for s in status if s == 0)} accepted by the reference -> {out} ({os.path.getsize(out)} bytes)")


if __name__ == "__main__":
    main()
