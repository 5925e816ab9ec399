"""Put this directory in front of a reference checkout on PYTHONPATH."""
