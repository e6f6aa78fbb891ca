"""Extract the raw fp64 buffers of the reference's only golden result file
WITHOUT unpickling it (opcode scan with pickletools.genops, which executes
nothing from the file), and store them as a plain .npz.

Source: agimus_controller/tests/resources/simple_ocp_croco_results.pkl, checked
by agimus_controller/tests/test_ocp_croco_base.py:175-204.  The pickle only
references numpy.core.numeric._frombuffer and numpy.dtype('<f8'); its byte
strings are: 10 states x 14, 9 Riccati gains x 98 (7x14, C order), 9 controls x 7.

Run here (needs /root/reference): python tests/golden/make_simple_ocp_fixture.py
"""
import pathlib
import pickletools

import numpy as np

SRC = pathlib.Path("/root/reference/agimus_controller/tests/resources/simple_ocp_croco_results.pkl")
DST = pathlib.Path(__file__).resolve().parent / "simple_ocp_croco_results.npz"


def main():
    bufs = []
    for op, arg, _ in pickletools.genops(SRC.read_bytes()):
        if op.name in ("BYTEARRAY8", "BINBYTES", "SHORT_BINBYTES", "BINBYTES8"):
            bufs.append(np.frombuffer(bytes(arg), dtype="<f8").copy())
    assert [b.size for b in bufs] == [14] * 10 + [98] * 9 + [7] * 9
    states = np.stack(bufs[:10])
    gains = np.stack(bufs[10:19]).reshape(9, 7, 14)
    controls = np.stack(bufs[19:])
    np.savez(DST, states=states, ricatti_gains=gains, feed_forward_terms=controls)
    print("wrote", DST, states.shape, gains.shape, controls.shape)


if __name__ == "__main__":
    main()
