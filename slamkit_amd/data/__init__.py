from .hf_dataset import (DataCollatorForLanguageModeling, DataCollatorWithFlattening, TokenDataset, TokenShardDataset,
                         chunk_texts, get_filter_fn, init_dataset, init_preference_optimization_dataset,
                         interleave_datasets, parse_single_dataset, split_into_chunks, write_token_shard)

__all__ = ["init_dataset", "init_preference_optimization_dataset", "chunk_texts", "split_into_chunks", "get_filter_fn",
           "parse_single_dataset", "TokenDataset", "TokenShardDataset", "write_token_shard", "interleave_datasets",
           "DataCollatorForLanguageModeling", "DataCollatorWithFlattening"]
