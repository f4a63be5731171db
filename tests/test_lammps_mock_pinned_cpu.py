"""CPU, build container only: pin the LAMMPS MOCK (tests/lammps_mock/) to LAMMPS itself.

The glue (lammps/pair_e3gnn_hip.cpp, pair_d3_hip.cpp) is compiled and executed against a mock this repository wrote, so a signature
that differs from LAMMPS would compile here and fail for the first user.  No LAMMPS tree exists in the image, but the reference
holds stock LAMMPS text and code written against the target release (stable_2Aug2023_update3): the vendored
sevenn/pair_e3gnn/comm_brick.{h,cpp} (LAMMPS' own CommBrick) and the reference's pair styles pair_e3gnn.cpp,
pair_e3gnn_parallel.cpp, pair_d3.cu.  This test extracts, at test time and from the files where they lie under /root/reference
(nothing is copied), every LAMMPS call and member access those sources make, and checks:
  * the mock declares CommBrick's entry points and the pair-side hooks with the argument lists stock LAMMPS calls them with;
  * every LAMMPS call / member the GLUE uses is one the reference's sources use too, with the same number of arguments --
    whatever is not is exactly the list INTEGRATION.md publishes as unverified.
Skipped where /root/reference is absent (the GPU box)."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = '/root/reference/sevenn/pair_e3gnn'
pytestmark = pytest.mark.skipif(not os.path.isdir(REF), reason='needs the reference tree (build container)')

OBJECTS = ('memory', 'error', 'atom', 'comm', 'neighbor', 'domain', 'force', 'list', 'utils', 'NeighConst')


def _strip(text):
    text = re.sub(r'/\*.*?\*/', ' ', text, flags=re.S)
    text = re.sub(r'//[^\n]*', ' ', text)
    return re.sub(r'"(?:\\.|[^"\\])*"', '""', text)


def _args(text, k):
    """text[k] == '(' -> number of top-level arguments of the call"""
    depth, n, seen = 0, 0, False
    for ch in text[k:]:
        if ch in '([{':
            depth += 1
        elif ch in ')]}':
            depth -= 1
            if depth == 0:
                return n + (1 if seen else 0)
        elif ch == ',' and depth == 1:
            n += 1
        elif depth >= 1 and not ch.isspace():
            seen = True
    raise AssertionError('unbalanced call')


def lammps_uses(text):
    """{(object, name): set of argument counts, or {None} for a data member} over the LAMMPS singletons a pair style reaches"""
    text = _strip(text)
    out = {}
    for m in re.finditer(r'\b(' + '|'.join(OBJECTS) + r')\s*(?:->|::)\s*(\w+)\s*(\()?', text):
        key = (m.group(1), m.group(2))
        out.setdefault(key, set()).add(_args(text, m.end() - 1) if m.group(3) else None)
    return out


def _read(*names, base=REF):
    return '\n'.join(open(os.path.join(base, n)).read() for n in names)


def test_comm_brick_entry_points_and_pair_hooks_match_stock_lammps():
    """stock LAMMPS (vendored comm_brick.{h,cpp}): Comm::forward_comm(Pair *) / reverse_comm(Pair *) and the four pair-side hooks
    with the arguments CommBrick passes them; the mock's Comm / Pair must declare exactly those"""
    h, c = _strip(_read('comm_brick.h')), _strip(_read('comm_brick.cpp'))
    assert re.search(r'void\s+forward_comm\s*\(\s*class\s+Pair\s*\*\s*\)\s*override', h)
    assert re.search(r'void\s+reverse_comm\s*\(\s*class\s+Pair\s*\*\s*\)\s*override', h)
    calls = {}
    for m in re.finditer(r'\bpair\s*->\s*(\w+)\s*\(', c):
        calls.setdefault(m.group(1), set()).add(_args(c, m.end() - 1))
    assert calls['pack_forward_comm'] == {5} and calls['unpack_forward_comm'] == {3}
    assert calls['pack_reverse_comm'] == {3} and calls['unpack_reverse_comm'] == {3}
    # argument TYPES at the stock call sites, from CommBrick's own member declarations (comm_brick.h): indexed once, an
    # `int *` member is an int and an `int **` member an int *; the buffers are double *
    def member_type(name):
        m = re.search(r'\b(int|double)\s*([^;]*?)(\*{0,2})\s*' + name + r'\b[^;]*;', h)
        if m is None:   # a local of the calling function (double *buf)
            m = re.search(r'\b(int|double)\s*(\*{0,2})\s*' + name + r'\s*;', c)
        assert m, name
        decl = re.search(r'(\*{0,2})\s*' + name + r'\b', m.group(0)).group(1)
        return m.group(1), len(decl)

    def arg_type(expr):
        expr = expr.strip()
        base, idx = re.match(r'(\w+)((?:\[[^\]]*\])*)$', expr).groups()
        t, stars = member_type(base)
        stars -= idx.count('[')
        return t + (' ' + '*' * stars if stars else '')
    sites = {fn: re.search(r'pair\s*->\s*' + fn + r'\s*\(([^;]*?)\)\s*;', c).group(1).split(',')
             for fn in ('pack_forward_comm', 'unpack_forward_comm', 'pack_reverse_comm', 'unpack_reverse_comm')}
    want = {fn: ', '.join(arg_type(a) for a in args) for fn, args in sites.items()}
    assert want == {'pack_forward_comm': 'int, int *, double *, int, int *', 'unpack_forward_comm': 'int, int, double *',
                    'pack_reverse_comm': 'int, int, double *', 'unpack_reverse_comm': 'int, int *, double *'}, want
    norm = lambda s_: re.sub(r'\s+', ' ', re.sub(r'\s*\*\s*', ' *', s_)).strip()  # noqa: E731
    mock = _strip(_read('pair.h', 'lmp_mock_core.h', base=os.path.join(ROOT, 'tests', 'lammps_mock')))
    for fn, sig in want.items():
        m = re.search(r'virtual\s+(?:int|void)\s+' + fn + r'\s*\(([^)]*)\)', mock)
        assert m and norm(m.group(1)) == norm(sig), (fn, m and m.group(1))
    assert re.search(r'virtual\s+void\s+forward_comm\s*\(\s*Pair\s*\*\s*\)', mock) and re.search(r'virtual\s+void\s+reverse_comm\s*\(\s*Pair\s*\*\s*\)', mock)
    # ... and the glue overrides them with the same lists
    glue_h = _strip(_read('pair_e3gnn_hip.h', base=os.path.join(ROOT, 'lammps')))
    for fn in ('pack_forward_comm', 'unpack_forward_comm'):
        m = re.search(r'\b(?:int|void)\s+' + fn + r'\s*\(([^)]*)\)\s*override', glue_h)
        assert m and norm(m.group(1)) == norm(want[fn]), (fn, m and m.group(1))


def _unverified_block():
    text = open(os.path.join(ROOT, 'INTEGRATION.md')).read()
    m = re.search(r'<!-- lammps-unverified:begin -->(.*?)<!-- lammps-unverified:end -->', text, flags=re.S)
    assert m, 'INTEGRATION.md lost its lammps-unverified block'
    return set(re.findall(r'`(\w+(?:->|::)\w+)`', m.group(1)))


def test_every_lammps_call_of_the_glue_is_one_the_reference_makes():
    ref = lammps_uses(_read('pair_e3gnn.cpp', 'pair_e3gnn_parallel.cpp', 'pair_d3.cu', 'pair_e3gnn.h', 'pair_e3gnn_parallel.h', 'pair_d3.h',
                            'comm_brick.cpp'))
    # stock Comm entry points declared in the vendored header count as LAMMPS text too (the reference itself calls its own overloads)
    for m in re.finditer(r'\bvoid\s+(\w+)\s*\(([^)]*)\)\s*override', _strip(_read('comm_brick.h'))):
        ref.setdefault(('comm', m.group(1)), set()).add(len([a for a in m.group(2).split(',') if a.strip()]))
    glue = lammps_uses(_read('pair_e3gnn_hip.cpp', 'pair_e3gnn_hip.h', 'pair_d3_hip.cpp', 'pair_d3_hip.h', base=os.path.join(ROOT, 'lammps')))
    assert len(glue) >= 20 and ('memory', 'create') in glue and ('neighbor', 'add_request') in glue
    unverified, arity = set(), []
    for key, counts in sorted(glue.items()):
        sep = '::' if key[0] in ('utils', 'NeighConst') else '->'
        name = f'{key[0]}{sep}{key[1]}'
        if key not in ref:
            unverified.add(name)
            continue
        extra = {c for c in counts if c is not None} - {c for c in ref[key] if c is not None}
        if key == ('utils', 'logmesg'):
            extra = set()                       # variadic ({fmt} arguments)
        if extra:
            arity.append((name, sorted(extra), sorted(c for c in ref[key] if c is not None)))
    assert arity == [], f'called with argument counts the reference never uses: {arity}'
    assert unverified == {u for u in _unverified_block() if not u.startswith('Pair::')}, (sorted(unverified), sorted(_unverified_block()))
    # the mock declares everything the glue reaches (it compiles against it: tests/test_lammps_glue_cpu.py); here: every name exists
    mock = _strip(_read('pair.h', 'lmp_mock_core.h', 'mpi.h', base=os.path.join(ROOT, 'tests', 'lammps_mock')))
    for (obj, name) in glue:
        assert re.search(r'\b' + name + r'\b', mock), (obj, name)


def test_pair_base_members_the_glue_uses_are_the_ones_the_reference_uses():
    """unqualified members inherited from LAMMPS' Pair (flags, accumulators, ev_setup): the mock declares them; the glue may use
    only those the reference's pair styles use too"""
    mock = _strip(_read('pair.h', base=os.path.join(ROOT, 'tests', 'lammps_mock')))
    body = mock[mock.index('class Pair'):]
    members = ['eng_vdwl', 'eng_coul', 'virial', 'eatom', 'vatom', 'comm_forward', 'comm_reverse', 'single_enable', 'restartinfo', 'one_coeff',
               'manybody_flag', 'no_virial_fdotr_compute', 'list', 'allocated', 'setflag', 'cutsq', 'map', 'eflag_either', 'eflag_global',
               'eflag_atom', 'vflag_either', 'vflag_global', 'vflag_atom', 'ev_setup', 'evflag', 'vflag_fdotr', 'maxeatom', 'maxvatom']
    for m in members:
        assert re.search(r'\b' + m + r'\b', body), m                 # the list above is what the mock's Pair declares
    ref = _strip(_read('pair_e3gnn.cpp', 'pair_e3gnn_parallel.cpp', 'pair_d3.cu'))
    glue = _strip(_read('pair_e3gnn_hip.cpp', 'pair_d3_hip.cpp', base=os.path.join(ROOT, 'lammps')))
    used = {m for m in members if re.search(r'(?<![\w>.])' + m + r'\b', glue)}
    assert {'eng_vdwl', 'virial', 'setflag', 'cutsq', 'ev_setup', 'evflag', 'vflag_fdotr'} <= used
    missing = {'Pair::' + m for m in used if not re.search(r'(?<![\w>.])' + m + r'\b', ref)}
    # (the reference adds its energy / virial unconditionally; the glue honours LAMMPS' global flags)
    assert missing == {u for u in _unverified_block() if u.startswith('Pair::')}, sorted(missing)
