"""Run the reference's converter (`/root/reference/assets/scripts/convert_safetensors.py`) unmodified, in this process.

The script was written against an older `safetensors` whose low-level `serialize_file` took
`{name: {"dtype", "shape", "data": bytes}}`; safetensors 0.8 (installed here) takes `{name: TensorSpec}`.  This wrapper
only adapts that one library call (same bytes, same header: the library still writes the file) and then executes the
script as `__main__` with the caller's argv — version sniffing, renames, transposes, `.half()`, key lower-casing and
metadata are the reference's own lines (convert_safetensors.py:28-80, 96-101).

    python tests/golden/run_reference_converter.py --input model.pth --output out/model.st
"""
import runpy
import sys

import numpy as np
import safetensors.torch as _st
from safetensors._safetensors_rust import TensorSpec

CONVERTER = "/root/reference/assets/scripts/convert_safetensors.py"
_new_serialize_file = _st.serialize_file


def _serialize_file_compat(tensor_dict, filename, metadata=None):
    keep, specs = [], {}
    for name, t in tensor_dict.items():
        if isinstance(t, dict):
            buf = np.frombuffer(t["data"], dtype=np.uint8)
            keep.append(buf)
            specs[name] = TensorSpec(dtype=t["dtype"], shape=list(t["shape"]), data_ptr=buf.ctypes.data, data_len=buf.size)
        else:
            specs[name] = t
    return _new_serialize_file(specs, filename, metadata=metadata)


if __name__ == "__main__":
    _st.serialize_file = _serialize_file_compat
    sys.argv = [CONVERTER] + sys.argv[1:]
    runpy.run_path(CONVERTER, run_name="__main__")
