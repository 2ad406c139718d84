"""lz4_amd -- MI355X-native LZ4 block codec behind pierrec/lz4's API surface.

Layout:
  csrc/      hand-written gfx950 HIP kernels + the C ABI (include/lz4hip.h)
  block.py   mirror of lz4.CompressBlock / CompressBlockHC / UncompressBlock
  frame.py   mirror of lz4.Writer / lz4.Reader / lz4.CompressingReader and the options
  device.py  device-resident batched entry points on torch tensors
  corpus.py  seeded synthetic corpora of the benchmark configs
  shard.py   block-range sharding across GPUs (one process per GPU)
"""
from ._lib import LIB_PATH, Lz4HipError, load  # noqa: F401
from .block import (Block1Mb, Block4Mb, Block64Kb, Block256Kb, ChecksumZero,  # noqa: F401
                    CompressBlock, CompressBlockBound, CompressBlockHC, Compressor,
                    CompressorHC, ErrInvalidSourceShortBuffer, Fast, Level1, Level2, Level3,
                    Level4, Level5, Level6, Level7, Level8, Level9, UncompressBlock,
                    UncompressBlockWithDict)
from .frame import (BlockChecksumOption, BlockSizeOption, ChecksumOption,  # noqa: F401
                    CompressingReader, CompressionLevelOption, ConcurrencyOption,
                    ErrInvalidBlockChecksum, ErrInvalidFrame, ErrInvalidFrameChecksum,
                    ErrInvalidHeaderChecksum, ErrOptionClosedOrError,
                    ErrOptionInvalidBlockSize, ErrOptionInvalidCompressionLevel,
                    ErrOptionNotApplicable, LegacyOption, OnBlockDoneOption, Reader,
                    SizeOption, Writer)
