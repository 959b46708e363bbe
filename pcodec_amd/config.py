"""Config objects mirroring pco_python/src/config.rs (ModeSpec, DeltaSpec, PagingSpec, ChunkConfig)
and progress.rs (Progress)."""
from dataclasses import dataclass, field

from . import _lib as G


@dataclass(frozen=True)
class ModeSpec:  # chunk_config.rs:13-51, pco_python/src/config.rs:4-46
    kind: int = G.MODE_AUTO
    f64: float = 0.0
    u64: int = 0

    @staticmethod
    def auto(): return ModeSpec(G.MODE_AUTO)

    @staticmethod
    def classic(): return ModeSpec(G.MODE_CLASSIC)

    @staticmethod
    def try_float_mult(base): return ModeSpec(G.MODE_TRY_FLOAT_MULT, f64=float(base))

    @staticmethod
    def try_float_quant(k): return ModeSpec(G.MODE_TRY_FLOAT_QUANT, u64=int(k))

    @staticmethod
    def try_int_mult(base): return ModeSpec(G.MODE_TRY_INT_MULT, u64=int(base))

    @staticmethod
    def try_dict(): return ModeSpec(G.MODE_TRY_DICT)


@dataclass(frozen=True)
class DeltaSpec:  # chunk_config.rs:61-109, pco_python/src/config.rs:48-84
    kind: int = G.DELTA_AUTO
    order: int = 0

    @staticmethod
    def auto(): return DeltaSpec(G.DELTA_AUTO)

    @staticmethod
    def no_op(): return DeltaSpec(G.DELTA_NOOP)

    @staticmethod
    def try_consecutive(order): return DeltaSpec(G.DELTA_TRY_CONSECUTIVE, int(order))

    @staticmethod
    def try_lookback(): return DeltaSpec(G.DELTA_TRY_LOOKBACK)

    @staticmethod
    def try_conv1(order): return DeltaSpec(G.DELTA_TRY_CONV1, int(order))


@dataclass(frozen=True)
class PagingSpec:  # chunk_config.rs:112-182, pco_python/src/config.rs:86-106
    max_page_n: int = 1 << 18
    exact: tuple = None

    @staticmethod
    def equal_pages_up_to(n): return PagingSpec(max_page_n=int(n))

    @staticmethod
    def exact_page_sizes(sizes): return PagingSpec(exact=tuple(int(s) for s in sizes))


@dataclass
class ChunkConfig:  # chunk_config.rs:191-235, pco_python/src/config.rs:108-159
    compression_level: int = 8
    mode_spec: ModeSpec = field(default_factory=ModeSpec.auto)
    delta_spec: DeltaSpec = field(default_factory=DeltaSpec.auto)
    paging_spec: PagingSpec = field(default_factory=PagingSpec)
    enable_8_bit: bool = False

    def to_c(self, wrapped=False):
        # (PagingSpec.exact travels beside the struct: pco_chunk_compressor_new_exact / pco_gfx_simple_compress_into_exact)
        return G.make_config(level=self.compression_level, mode=self.mode_spec.kind, mode_f64=self.mode_spec.f64,
                             mode_u64=self.mode_spec.u64, delta=self.delta_spec.kind, delta_order=self.delta_spec.order,
                             max_page_n=self.paging_spec.max_page_n, enable_8_bit=self.enable_8_bit)


@dataclass
class Progress:  # progress.rs:3-12
    n_processed: int = 0
    finished: bool = False
