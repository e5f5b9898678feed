"""hippopt_amd — MI355X-native multiple-shooting NLP-callback engine behind hippopt's
solver boundary."""
