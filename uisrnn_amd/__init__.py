"""MI355X-native UIS-RNN beam-search decoder (drop-in for uisrnn.UISRNN.predict)."""
