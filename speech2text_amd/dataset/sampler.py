"""Dynamic bucket batch sampler (host-side mirror of the reference's dataset/sampler.py:17-96):
samples are routed to `num_bucket` duration buckets between the dataset's bounds; a bucket is
emitted as a batch when its accumulated volume exceeds `volume_threshold` seconds and it holds
more than `min_batch_size` samples; the underlying sampler is replayed endlessly."""
import math

from torch.utils.data.sampler import BatchSampler


class DynamicBucketBatchSampler(BatchSampler):
    def __init__(self, sampler, dataset, num_bucket: int = 30, key: str = "duration",
                 min_batch_size: int = 8, volume_threshold: int = 800) -> None:
        super().__init__(sampler, min_batch_size, drop_last=False)
        assert hasattr(dataset, "fetch_data_k_info")
        self._key = key
        self._dataset = dataset
        self._volume_threshold = volume_threshold
        lo, hi = float(dataset.lower_bound), float(dataset.high_bound)
        step = (hi - lo) / float(num_bucket)
        self._buckets = {i: {"bucket_id": i, "data": [], "bounds": (i * step + lo, (i + 1) * step + lo),
                             "volume": 0.0} for i in range(num_bucket)}

    def _select_bucket(self, v):
        for i, b in self._buckets.items():
            if b["bounds"][0] <= v <= b["bounds"][1]:
                return i
        return None

    def __iter__(self):
        while True:
            for sample_id in self.sampler:
                v = self._dataset.fetch_data_k_info(sample_id, k=self._key)
                b = self._buckets[self._select_bucket(v)]
                b["data"].append(sample_id)
                b["volume"] += v
                if b["volume"] > self._volume_threshold and len(b["data"]) > self.batch_size:
                    yield b["data"]
                    b["data"] = []
                    b["volume"] = 0.0

    def __len__(self) -> int:
        return math.ceil(math.ceil(self._dataset.total_data_amount / self.sampler.num_replicas)
                         / self._volume_threshold)
