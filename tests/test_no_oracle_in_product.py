"""The oracle is test infrastructure: nothing under textgcn_amd/ may import, load or execute it, and the
package must not read /root/reference."""
import os
import re

from conftest import ROOT


def product_sources():
    for base, _, files in os.walk(os.path.join(ROOT, 'textgcn_amd')):
        for f in files:
            if f.endswith(('.py', '.hip', '.h', '.cpp')):
                yield os.path.join(base, f)


def test_product_never_touches_oracle_or_reference():
    for path in product_sources():
        src = open(path).read()
        assert not re.search(r'^\s*(from|import)\s+oracle\b', src, flags=re.M), path
        assert 'lgcn_oracle' not in src, path
        assert '/root/reference' not in src, path
        assert 'import TextGCN' not in src, path


def test_bench_and_entry_do_not_read_reference():
    for f in ('bench.py', '__graft_entry__.py'):
        p = os.path.join(ROOT, f)
        if os.path.exists(p):
            src = open(p).read()
            assert 'import TextGCN' not in src and 'sys.path.insert(0, \'/root/reference\')' not in src
