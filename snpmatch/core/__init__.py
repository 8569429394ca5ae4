"""``snpmatch.core.X`` IS ``snpmatch_amd.core.X`` (same module object, so module constants such as
``lr_thres`` stay mutable through either name)."""
import importlib
import sys

for _name in ("parsers", "genomes", "snp_genotype", "snpmatch", "csmatch"):
    _mod = importlib.import_module("snpmatch_amd.core." + _name)
    sys.modules[__name__ + "." + _name] = _mod
    globals()[_name] = _mod
