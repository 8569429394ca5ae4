"""
Drop-in import path: ``snpmatch.core.snpmatch`` / ``snpmatch.core.csmatch`` (and parsers,
snp_genotype, genomes) resolve to the MI355X implementation in ``snpmatch_amd.core``; ``snpmatch.main``
is the CLI with the ``inbred`` and ``cross`` subcommands.
"""
from snpmatch_amd.cli import main, get_options, setLog  # noqa: F401
from snpmatch_amd.cli import __version__  # noqa: F401
