"""Regenerates the ctypes block of INTEGRATION.md section 3 (between the BEGIN/END GENERATED markers) from casapose_amd/_lib.py, so the
documented `ConvSource` / `ConvDesc` declarations cannot drift from the binding that is tested against the header
(tests/test_capi_symbols.py::test_integration_doc_binding_is_current).   python tools/gen_integration_binding.py [--check]"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
BEGIN, END = "<!-- BEGIN GENERATED: ctypes binding (tools/gen_integration_binding.py) -->", "<!-- END GENERATED -->"


def _tname(t) -> str:
    from casapose_amd import _lib

    if isinstance(t, type) and issubclass(t, C.Array):
        return "%s * %d" % (_tname(t._type_), t._length_)
    if t is _lib.ConvSource:
        return "ConvSource"
    return "C.c_uint32" if t is C.c_uint32 else "C." + t.__name__


def _fields(cls, indent: str) -> str:
    items = ['("%s", %s)' % (n, _tname(t)) for n, t in cls._fields_]
    lines, cur = [], indent
    for i, it in enumerate(items):          # whole ("name", type) pairs per line
        piece = it + ("," if i + 1 < len(items) else "")
        if len(cur) + len(piece) + 1 > 118 and cur.strip():
            lines.append(cur.rstrip())
            cur = indent
        cur += piece + " "
    lines.append(cur.rstrip())
    return "\n".join(lines)


def block() -> str:
    from casapose_amd import _lib

    lines = [
        "```python",
        "import ctypes as C",
        'lib = C.CDLL("casapose_amd/libcasapose_hip.so")',
        "assert lib.cp_version() == %d                       # CP_ABI_VERSION of include/casapose_hip.h" % _lib.ABI_VERSION,
        "",
        "class ConvSource(C.Structure):            # cp_conv_source",
        "    _fields_ = [",
        _fields(_lib.ConvSource, "        "),
        "    ]",
        "",
        "class ConvDesc(C.Structure):              # cp_conv_desc -- field order as in the header; struct_size = sizeof, checked by the library",
        "    _fields_ = [",
        _fields(_lib.ConvDesc, "        "),
        "    ]",
        "    def __init__(self, *a, **k):",
        "        super().__init__(*a, **k)",
        "        self.struct_size = C.sizeof(ConvDesc)",
        "",
        "lib.cp_conv_desc_size.restype = C.c_size_t",
        "assert lib.cp_conv_desc_size() == C.sizeof(ConvDesc)                  # a stale declaration is refused here, not inside a kernel",
        "lib.cp_conv2d_fwd_f32.argtypes = [C.POINTER(ConvDesc), C.c_void_p]     # (desc, hipStream_t)",
        "lib.cp_conv2d_fwd_f32.restype = C.c_int                                # 0 ok, <0 -> cp_last_error()",
        "lib.cp_ls_vote_f32.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p,",
        "                               C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]",
        "lib.cp_last_error.restype = C.c_char_p",
        "```",
    ]
    return "\n".join(lines)


def main():
    path = os.path.join(ROOT, "INTEGRATION.md")
    text = open(path).read()
    a, b = text.index(BEGIN), text.index(END)
    new = text[:a] + BEGIN + "\n" + block() + "\n" + text[b:]
    if "--check" in sys.argv:
        sys.exit(0 if new == text else 1)
    open(path, "w").write(new)


if __name__ == "__main__":
    main()
