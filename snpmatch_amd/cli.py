"""
Command line: ``snpmatch inbred`` and ``snpmatch cross`` with the reference's flags
(snpmatch/__init__.py:44-63), logging setup (:23-34) and exit codes (:155-183), plus ``makedb-native``
to write the flat panel format this engine streams to the GPU.  The other reference subcommands
(genotype_cross, parser, pairsnp, makedb, simulate) are outside the accelerated path.
"""
import argparse
import logging
import os
import sys

__version__ = "0.1.0 (SNPmatch 5.0.1 inbred/cross interface)"


def setLog(logDebug):
    log = logging.getLogger()
    numeric_level = logging.DEBUG if logDebug else logging.ERROR
    log_format = logging.Formatter("%(asctime)s - %(name)s - %(levelname)s - %(message)s")
    lch = logging.StreamHandler()
    lch.setLevel(numeric_level)
    lch.setFormatter(log_format)
    log.setLevel(numeric_level)
    log.addHandler(lch)


def die(msg):
    sys.stderr.write('Error: ' + msg + '\n')
    sys.exit(1)


def check_file(inFile):
    if not inFile:
        die("file: %s not specified" % inFile)
    if not os.path.exists(inFile):
        die("input file does not exist: " + inFile)


def snpmatch_inbred(args):
    from .core import snpmatch
    check_file(args['inFile'])
    snpmatch.potatoGenotyper(args)


def snpmatch_inbred_batch(args):
    from .core import snpmatch
    for f in args['inFiles']:
        check_file(f)
    snpmatch.potatoGenotyperBatch(args)


def snpmatch_cross(args):
    from .core import csmatch
    check_file(args['inFile'])
    csmatch.potatoCrossIdentifier(args)


def makedb_native(args):
    """<db>.npz (snps, accessions, positions, chrs, chr_regions) or HDF5 -> <out>.snpm flat panel"""
    from .core import snp_genotype
    g = snp_genotype._load_any(args['inFile'])
    snp_genotype.save_native(args['outFile'], g.snps, g.accessions, g.positions, g.chrs, g.chr_regions, packed=args.get('packed', False))


def get_options(description, version_message):
    p = argparse.ArgumentParser(description=description)
    p.add_argument('-V', '--version', action='version', version=version_message)
    sub = p.add_subparsers(title='subcommands', description='Choose a command to run', help='Following commands are supported')

    def common(sp, default_out):
        sp.add_argument("-i", "--input_file", dest="inFile", help="VCF/BED file for the variants in the sample")
        sp.add_argument("-d", "--hdf5_file", default=None, dest="hdf5File",
                        help="Path to SNP matrix: native flat panel directory (.snpm), .npz, or HDF5 chunked row-wise")
        sp.add_argument("-e", "--hdf5_acc_file", default=None, dest="hdf5accFile",
                        help="Path to SNP matrix chunked column-wise (optional for flat panels)")
        sp.add_argument("--skip_db_hets", action="store_true", dest="skip_db_hets", default=False,
                        help="Replace heterozygous calls in DB with nan during the analysis.")
        sp.add_argument("-v", "--verbose", action="store_true", dest="logDebug", default=False, help="Show verbose debugging output")
        sp.add_argument("-o", "--output", dest="outFile", default=default_out, help="Output file prefix")

    inbred = sub.add_parser('inbred', help="SNPmatch on the inbred samples")
    common(inbred, "identify_inbred")
    inbred.add_argument("--refine", action="store_true", dest="refine", default=False, help="Refine scores for indistinguishable lines")
    inbred.set_defaults(func=snpmatch_inbred)

    # not in the reference (which starts a process, and opens the DB, per sample): many samples against ONE resident DB
    batch = sub.add_parser('inbred-batch', help="`inbred` for many samples: the DB is loaded once, samples are scored a batch per device call")
    batch.add_argument("-i", "--input_files", dest="inFiles", nargs='+', required=True, help="VCF/BED files, one sample each")
    batch.add_argument("-d", "--hdf5_file", default=None, dest="hdf5File", help="Path to SNP matrix (as for inbred)")
    batch.add_argument("-e", "--hdf5_acc_file", default=None, dest="hdf5accFile", help="Path to SNP matrix chunked column-wise (optional for flat panels)")
    batch.add_argument("--skip_db_hets", action="store_true", dest="skip_db_hets", default=False,
                       help="Replace heterozygous calls in DB with nan during the analysis.")
    batch.add_argument("-v", "--verbose", action="store_true", dest="logDebug", default=False, help="Show verbose debugging output")
    batch.add_argument("-o", "--output", dest="outFile", default="identify_inbred",
                       help="Output prefix: sample <name>.vcf writes <prefix>.<name>.scores.txt / .matches.json")
    batch.add_argument("--batch_size", dest="batchSize", default=64, type=int, help="samples per device call")
    batch.set_defaults(func=snpmatch_inbred_batch)

    cross = sub.add_parser('cross', help="SNPmatch on the crosses (F2s and F3s) of A. thaliana")
    common(cross, "identify_cross")
    cross.add_argument("-b", "--binLength", dest="binLen", help="Length of bins to calculate the likelihoods", default=300000, type=int)
    cross.add_argument("--genome", dest="genome", default="athaliana_tair10",
                       help="Genome id or path to a reference JSON file (ref_chrs, ref_chrlen)")
    cross.set_defaults(func=snpmatch_cross)

    mk = sub.add_parser('makedb-native', help="Convert a DB (.npz / HDF5) to the native flat panel format")
    mk.add_argument("-i", "--input", dest="inFile")
    mk.add_argument("-o", "--output", dest="outFile")
    mk.add_argument("--packed", action="store_true", dest="packed", default=False,
                    help="store 2 bits per call (a quarter of the disk and of the bytes a load moves; DBs with the codes -1/0/1/2 only)")
    mk.add_argument("-v", "--verbose", action="store_true", dest="logDebug", default=False)
    mk.set_defaults(func=makedb_native)
    return p


def main(argv=None):
    parser = get_options("SNPmatch inbred / cross scoring on MI355X (snpmatch_amd)", '%(prog)s ' + __version__)
    args = vars(parser.parse_args(argv))
    setLog(args.get('logDebug', False))
    if 'func' not in args:
        parser.print_help()
        return 0
    job = None
    if int(os.environ.get("WORLD_SIZE", "1") or 1) <= 1:
        # a plain CLI run never imports torch (several GPUs are driven through the library's own RCCL group): stay on the
        # HIP runtime the library was built with instead of the copy bundled with a PyTorch wheel
        os.environ.setdefault("SNPMATCH_HIP_RUNTIME", "system")
    try:
        from . import dist
        job = dist.init_from_env()       # under torch.distributed.run: accession-sharded over the ranks' GPUs
        args['func'](args)
        if job is not None:
            import torch.distributed as td
            td.barrier()
            td.destroy_process_group()
        return 0
    except KeyboardInterrupt:
        return 0
    except Exception as e:
        logging.exception(e)
        if job is not None:
            # a rank that fails must not leave its peers waiting in a collective until the launcher's timeout: tear the
            # communicator down (abort where the backend has it) so that their pending calls fail at once
            try:
                import torch.distributed as td
                pg = td.distributed_c10d._get_default_group()
                if hasattr(pg, "abort"):
                    pg.abort()
                else:
                    td.destroy_process_group()
            except Exception:
                pass
        return 2


if __name__ == '__main__':
    sys.exit(main())
