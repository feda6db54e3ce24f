"""The three places that know the translation units of libecrad_hip.so -- ecrad_amd/csrc/Makefile and the variant builders under tools/
(which build the libraries tests/test_hip_parity.py compares with the shipped one) -- list the same files, every .hip file of the
directory is in the list, and the per-file compiler flags of the Makefile are the ones the variant builders repeat."""
import glob
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "ecrad_amd", "csrc")


def _makefile():
    return open(os.path.join(CSRC, "Makefile")).read()


def _makefile_sources():
    m = re.search(r"^SRC\s*=\s*(.*)$", _makefile(), re.M)
    return sorted(f[:-4] for f in m.group(1).split())


def _script_sources(name):
    text = open(os.path.join(ROOT, "tools", name)).read()
    m = re.search(r'(?:src="|for o in )((?:pool|setup)[^";]*)', text)
    return sorted(m.group(1).split())


def test_every_hip_file_is_built():
    on_disk = sorted(os.path.basename(f)[:-4] for f in glob.glob(os.path.join(CSRC, "*.hip")))
    assert _makefile_sources() == on_disk


def test_variant_builders_list_the_same_translation_units():
    want = _makefile_sources()
    for script in ("variants.sh", "variant_files.sh", "variant_flags.sh"):
        assert _script_sources(script) == want, script


def test_variant_builders_repeat_the_per_file_flags():
    mk = _makefile()
    extras = dict(re.findall(r"^EXTRA_(\w+)\s*=\s*(.*)$", mk, re.M))
    assert set(extras) == {"kernel_ica_lw_clear", "kernel_spartacus", "kernel_spartacus_lw"}
    fast_div = re.search(r"^SP_FAST_DIV\s*=\s*(.*)$", mk, re.M).group(1).strip()
    assert "-fno-slp-vectorize" in extras["kernel_spartacus"] and "-fno-slp-vectorize" not in extras["kernel_spartacus_lw"]
    for script in ("variants.sh", "variant_files.sh"):
        text = open(os.path.join(ROOT, "tools", script)).read()
        assert extras["kernel_ica_lw_clear"].strip() in text, script
        assert fast_div in text and "-fno-slp-vectorize" in text, script
        # the nopack variant of the tests asks for the correctly rounded division everywhere
        assert "FAST_DIV=0" in text, script
