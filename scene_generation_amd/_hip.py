"""ctypes binding of libsg2im_hip.so.

The prototypes are parsed from include/sg2im_hip.h (the single source of truth for the C ABI), so the
Python argtypes can never drift from the header.  There is NO fallback: if the shared library is
missing or a symbol cannot be resolved, import of the compute path fails loudly.
"""
import ctypes
import os
import re

_HERE = os.path.dirname(os.path.abspath(__file__))
HEADER = os.path.join(os.path.dirname(_HERE), 'include', 'sg2im_hip.h')
LIB_PATH = os.environ.get('SG_LIB_PATH') or os.path.join(_HERE, 'csrc', 'libsg2im_hip.so')   # SG_LIB_PATH: kernel-variant builds


class sgConvDesc(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int32) for n in
                ('N', 'C1', 'C2', 'H', 'W', 'Cout', 'KS', 'stride', 'pad', 'pad_reflect', 'upsample', 'OH', 'OW',
                 'out_pad', 'x2_broadcast')]


_SCALARS = {'int': ctypes.c_int, 'int32_t': ctypes.c_int32, 'int64_t': ctypes.c_int64, 'float': ctypes.c_float,
            'size_t': ctypes.c_size_t, 'double': ctypes.c_double, 'sgStream': ctypes.c_void_p}


def _ctype(decl):
    decl = decl.strip()
    if decl == 'void':
        return None
    if '*' in decl:
        if 'char' in decl:
            return ctypes.c_char_p
        if decl.replace('const', '').strip().startswith('double'):
            return ctypes.POINTER(ctypes.c_double)
        if re.match(r'(const\s+)?int64_t\s*\*\s*(launches)?$', decl) and 'launches' in decl:
            return ctypes.POINTER(ctypes.c_int64)
        return ctypes.c_void_p
    base = decl.replace('const', '').split()[0]
    return _SCALARS[base]


def parse_header(path=HEADER):
    """-> {name: (restype, [argtypes], [argnames])} for every function the header declares."""
    src = open(path).read()
    src = re.sub(r'/\*.*?\*/', ' ', src, flags=re.S)
    src = re.sub(r'//[^\n]*', ' ', src)
    src = re.sub(r'typedef\s+struct\s+\w+\s*\{.*?\}\s*\w+\s*;', ' ', src, flags=re.S)
    src = re.sub(r'enum\s*\{.*?\}\s*;', ' ', src, flags=re.S)
    protos = {}
    for m in re.finditer(r'([A-Za-z_][\w\s\*]*?)\b(sg_\w+)\s*\(([^)]*)\)\s*;', src):
        ret, name, args = m.group(1).strip(), m.group(2), m.group(3).strip()
        if 'char' in ret:
            restype = ctypes.c_char_p
        elif ret == 'size_t':
            restype = ctypes.c_size_t
        else:
            restype = ctypes.c_int
        argtypes, argnames = [], []
        if args and args != 'void':
            for a in args.split(','):
                a = a.strip()
                nm = re.search(r'(\w+)\s*$', a).group(1)
                ty = a[:a.rfind(nm)].strip()
                if nm == 'launches':
                    argtypes.append(ctypes.POINTER(ctypes.c_int64))
                else:
                    argtypes.append(_ctype(ty))
                argnames.append(nm)
        protos[name] = (restype, argtypes, argnames)
    return protos


PROTOS = parse_header()
_lib = None


class HipLibraryMissing(RuntimeError):
    pass


def lib():
    """Load (once) and return the shared library with typed prototypes.  Raises if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.isfile(LIB_PATH):
        raise HipLibraryMissing(
            '%s not found: the MI355X compute path has no fallback. Build it with '
            '`python -c "import __graft_entry__ as g; g.build()"` or scene_generation_amd/csrc/build.sh' % LIB_PATH)
    L = ctypes.CDLL(LIB_PATH, mode=ctypes.RTLD_GLOBAL)
    for name, (restype, argtypes, _) in PROTOS.items():
        fn = getattr(L, name)           # AttributeError if the .so does not export a declared symbol
        fn.restype = restype
        fn.argtypes = argtypes
    _lib = L
    if os.environ.get('SG_LEGACY_ALIGN_CORNERS', '0') == '1':
        L.sg_set_legacy_align_corners(1)
    return L


def last_error():
    return lib().sg_last_error_string().decode()


def check(rc, name):
    if rc != 0:
        raise RuntimeError('%s failed (rc=%d): %s' % (name, rc, last_error()))


# ---- tuning / debugging switches of the library (include/sg2im_hip.h: sg_set_option) --------------------------------
OPTION_GENERATION = 0


def set_option(name, value):
    """Change a launch-plan switch of the library at run time (they are otherwise fixed when the library is loaded: compiled-in
    default, or the environment variable SG_<NAME>)."""
    check(lib().sg_set_option(name.encode(), int(value)), 'sg_set_option')
    global OPTION_GENERATION
    OPTION_GENERATION += 1          # shape-only queries memoised on the conv descs (ops/_core._q) may depend on an option


_opt_cache = {}


def option_cached(name):
    """current value of a switch, re-read from the library only after a set_option (per-launch callers)"""
    hit = _opt_cache.get(name)
    if hit is None or hit[0] != OPTION_GENERATION:
        hit = _opt_cache[name] = (OPTION_GENERATION, get_option(name))
    return hit[1]


def get_option(name):
    v = ctypes.c_int(0)
    check(lib().sg_get_option(name.encode(), ctypes.cast(ctypes.pointer(v), ctypes.c_void_p)), 'sg_get_option')
    return v.value


def options():
    """{name: (current value, compiled-in default)} of every switch"""
    L = lib()
    out = {}
    for i in range(L.sg_num_options()):
        name = L.sg_option_name(i).decode()
        out[name] = (get_option(name), L.sg_option_default(i))
    return out
